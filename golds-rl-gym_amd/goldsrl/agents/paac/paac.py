"""GridPAACLearner (reference fed_gym/agents/paac/paac.py:216-419) on the device engine.

The while-loop body -- T x (predict, sample, transform, step, bookkeeping), bootstrap, n-step returns,
train_step -- is two C calls (grl_net_rollout, grl_net_train_rollout); this class keeps the reference's
constructor, counters, lr anneal and log line."""
import logging
import time

from ... import _ffi
from ... import distributed as D
from ... import rollout as R
from .actor_learner import ActorLearner


class _Update(R._GradientExchange):
    """rollout.py's gradient step (RCCL inside train_rollout, or the host exchange) on the learner's bound net."""

    def __init__(self, net):
        self.net, self.lr = net, 0.0

    def step(self, lr):
        self.lr = lr
        return self._update()


class PAACLearner(ActorLearner):
    def __init__(self, network_creator, environment_creator, args, emulator_class, state_processor):
        super(PAACLearner, self).__init__(network_creator, environment_creator, args, emulator_class)
        self.workers = args.emulator_workers
        self.rnn_length = args.rnn_length
        self.state_processor = state_processor

    def train(self, max_updates=None):
        """PAACLearner.train (paac.py:55-209) for SolowRunner: masked returns, rewards clipped to +-2."""
        device_id = 0
        if isinstance(self.device, str) and ':' in self.device:
            device_id = int(self.device.rsplit(':', 1)[1])
        p, q = getattr(self.environment_creator, "p", 1), getattr(self.environment_creator, "q", 1)
        ranks = self._ranks()
        device_id = ranks.local_rank if ranks.world > 1 else device_id
        self.engine = _ffi.Engine(_ffi.ENV_SOLOW, self.emulator_counts, device_id=device_id, solow_p=p, solow_q=q,
                                  rnn_length=self.rnn_length, max_episode_steps=int(getattr(self, "max_episode_steps", 1024)),
                                  seed=int(getattr(self, "seed", 1692)), env_id_offset=ranks.rank * self.emulator_counts)
        self.engine.reset()
        self.network.bind(self.engine, rnn_length=self.rnn_length, gamma=self.gamma,
                          max_samples=self.emulator_counts * self.max_local_steps)
        net = self.network.net
        if self.resume:
            self.load_checkpoint(self.resume)
        return self._loop(net, ranks, lambda: net.rollout(self.max_local_steps), max_updates)

    def _ranks(self):
        """One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK from the launcher): this rank owns `emulator_counts` envs with
        global ids rank*emulator_counts.. (the np.split of runners.py:18-19 across GPUs instead of worker processes)."""
        if getattr(self, "ranks", None) is None:
            self.ranks = D.Ranks().init()
            if self.ranks.world > 1:      # before the engine's first HIP call: the runtime's threads inherit the mask (goldsrl/affinity.py)
                from ... import affinity
                self.cpu_affinity = affinity.pin_to_gpu(self.ranks.local_rank)
        return self.ranks

    def _loop(self, net, ranks, do_rollout, max_updates, between_updates=None):
        """The while-loop of PAACLearner.train / GridPAACLearner.train (paac.py:119-209, 302-406): rollout, bookkeeping (R6),
        gradient step [exchange over ranks], log line."""
        upd = _Update(net)
        self.gradient_exchange = D.attach_gradient_exchange(upd, ranks)
        world, E, T = ranks.world, self.emulator_counts, self.max_local_steps
        self.engine.episodes_enable(capacity=T * E)
        self._ep_steps_seen = 0
        self._open_summaries()
        logging.debug("Starting training at Step {}".format(self.global_step))
        counter, global_step_start, start_time = 0, self.global_step, time.time()
        stats = None
        while self.global_step < self.max_global_steps:
            loop_start_time = time.time()
            do_rollout()
            self._account_episodes(ranks)
            self.global_step += T * E * world      # global_step += 1 per env per step (paac.py:149, 341)
            stats = upd.step(self.get_lr())
            self._log_update(stats)
            counter += 1
            if between_updates is not None:
                between_updates(counter)
            if counter % max(1, int(5048 / (E * world))) == 0 and ranks.rank == 0:
                curr_time = time.time()
                last_ten = 0.0 if len(self.total_rewards) < 1 else sum(self.total_rewards[-10:]) / len(self.total_rewards[-10:])
                logging.info("Ran {} steps, at {} steps/s ({} steps/s avg), last 10 rewards avg {}"
                             .format(self.global_step,
                                     T * E * world / (curr_time - loop_start_time),
                                     (self.global_step - global_step_start) / (curr_time - start_time),
                                     last_ten))
            if max_updates is not None and counter >= max_updates:
                break
        return stats

    def _account_episodes(self, ranks):
        """R6 (paac.py:142-157, 331-349): the device kept total_episode_rewards / emulator_steps per env while the rollout ran;
        here the finished episodes become `rl/reward` summaries at the reference's global_step (global_step += 1 per env in
        env order inside step t) and entries of total_rewards (reward per step of the episode)."""
        E, T, world = self.emulator_counts, self.max_local_steps, ranks.world
        recs = self.engine.episodes_read()
        w = self.summary_writer
        for r in recs:
            t_in = int(r["step_index"]) - 1 - self._ep_steps_seen
            gstep = self.global_step + t_in * E * world + ranks.rank * E + int(r["env"]) + 1
            self.total_rewards.append(float(r["total_reward"]) / int(r["length"]))
            self.episode_log.append((gstep, ranks.rank * E + int(r["env"]), int(r["length"]), float(r["total_reward"])))
            if w is not None:
                w.add_scalar("rl/reward", float(r["total_reward"]), gstep)
        if w is not None and len(recs):
            w.flush()
        del self.total_rewards[:-1000]      # the log line reads the last ten only
        del self.episode_log[:-100000]
        self._ep_steps_seen += T


class GridPAACLearner(PAACLearner):
    N_AGENTS = 10

    def __init__(self, network_creator, environment_creator, args, emulator_class, state_processor):
        super().__init__(network_creator, environment_creator, args, emulator_class, state_processor)
        self.real_batch_size = self.emulator_counts * self.N_AGENTS
        self.reward_layout = getattr(args, "reward_layout", "broadcast")    # 'reference' reproduces quirk Q4
        self.engine = None

    def rescale_reward(self, reward, lb=-2, ub=2):
        return reward        # paac.py:223-224: Swarm rewards are not clipped

    def train(self, max_updates=None):
        device_id = 0
        if isinstance(self.device, str) and ':' in self.device:
            device_id = int(self.device.rsplit(':', 1)[1])
        ranks = self._ranks()
        device_id = ranks.local_rank if ranks.world > 1 else device_id
        self.engine = _ffi.Engine(_ffi.ENV_SWARM, self.emulator_counts, device_id=device_id, grid_size=self.network.height,
                                  seed=int(getattr(self, "seed", 1692)), max_episode_steps=int(getattr(self, "max_episode_steps", 128)),
                                  env_id_offset=ranks.rank * self.emulator_counts)
        self.engine.reset()
        self.network.bind(self.engine, gamma=self.gamma)
        net = self.network.net
        if self.resume:
            self.load_checkpoint(self.resume)
        layout = 1 if self.reward_layout == 'reference' else 0
        # paac.py:229-236,277-282: SwarmPolicyMonitor on Swarm-eval-v0, evaluated every 30 s.  The reference runs it in a
        # thread; handles here are single-threaded, so the episode (its own 1-env handle) is played between two updates.
        pe, last_eval = None, time.time()
        eval_every = float(getattr(self, "eval_every", 0.0) or 0.0)
        if eval_every > 0 and ranks.rank == 0:      # only rank 0 evaluates: the other ranks build neither the monitor nor its env
            from ...envs import make
            from ..state_processors import SwarmStateProcessor
            from .policy_monitor import ScalarWriter, SwarmPolicyMonitor
            pe = SwarmPolicyMonitor(env=make("Swarm-eval-v0"), global_policy_net=self.network,
                                    state_processor=SwarmStateProcessor(grid_size=self.network.height),
                                    summary_writer=self._open_summaries() or ScalarWriter(self.debugging_folder), saver=None,
                                    network_conf=self.network.conf,
                                    learner=self)
        state = {"last_eval": last_eval}

        def between_updates(counter):
            if pe is not None and ranks.rank == 0 and time.time() - state["last_eval"] >= eval_every:
                pe.eval_once(max_sequence_length=self.rnn_length)
                state["last_eval"] = time.time()
            ckpt_every = int(getattr(self, "checkpoint_every", 0) or 0)
            if ckpt_every > 0 and counter % ckpt_every == 0 and ranks.rank == 0:
                self.save_checkpoint(getattr(self, "checkpoint_path", "checkpoint.npz"))
        return self._loop(net, ranks, lambda: net.rollout(self.max_local_steps, layout), max_updates, between_updates)

    def cleanup(self):
        if self.network.net is not None:
            self.network.net.close()
