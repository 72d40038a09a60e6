"""GridPAACLearner (reference fed_gym/agents/paac/paac.py:216-419) on the device engine.

The while-loop body -- T x (predict, sample, transform, step, bookkeeping), bootstrap, n-step returns,
train_step -- is two C calls (grl_net_rollout, grl_net_train_rollout); this class keeps the reference's
constructor, counters, lr anneal and log line."""
import logging
import time

from ... import _ffi
from .actor_learner import ActorLearner


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
        self.engine = _ffi.Engine(_ffi.ENV_SOLOW, self.emulator_counts, device_id=device_id, solow_p=p, solow_q=q,
                                  rnn_length=self.rnn_length, max_episode_steps=1024, seed=int(getattr(self, "seed", 1692)))
        self.engine.reset()
        self.network.bind(self.engine, rnn_length=self.rnn_length, gamma=self.gamma,
                          max_samples=self.emulator_counts * self.max_local_steps)
        net = self.network.net
        if self.resume:
            self.load_checkpoint(self.resume)
        self._open_summaries()
        counter, global_step_start, start_time = 0, self.global_step, time.time()
        stats = None
        while self.global_step < self.max_global_steps:
            loop_start_time = time.time()
            net.rollout(self.max_local_steps)
            self.global_step += self.max_local_steps * self.emulator_counts      # paac.py:149
            stats = net.train_rollout(self.get_lr())
            self._log_update(stats)
            counter += 1
            if counter % max(1, int(5048 / self.emulator_counts)) == 0:
                curr_time = time.time()
                logging.info("Ran {} steps, at {} steps/s ({} steps/s avg), loss {}"
                             .format(self.global_step,
                                     self.max_local_steps * self.emulator_counts / (curr_time - loop_start_time),
                                     (self.global_step - global_step_start) / (curr_time - start_time), stats["loss"]))
            if max_updates is not None and counter >= max_updates:
                break
        return stats


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
        self.engine = _ffi.Engine(_ffi.ENV_SWARM, self.emulator_counts, device_id=device_id, grid_size=self.network.height,
                                  seed=int(getattr(self, "seed", 1692)), max_episode_steps=128)
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
        if eval_every > 0:
            from ...envs import make
            from ..state_processors import SwarmStateProcessor
            from .policy_monitor import ScalarWriter, SwarmPolicyMonitor
            pe = SwarmPolicyMonitor(env=make("Swarm-eval-v0"), global_policy_net=self.network,
                                    state_processor=SwarmStateProcessor(grid_size=self.network.height),
                                    summary_writer=self._open_summaries() or ScalarWriter(self.debugging_folder), saver=None,
                                    network_conf=self.network.conf,
                                    learner=self)
        self._open_summaries()
        logging.debug("Starting training at Step {}".format(self.global_step))
        counter, global_step_start, start_time = 0, self.global_step, time.time()
        stats = None
        while self.global_step < self.max_global_steps:
            loop_start_time = time.time()
            net.rollout(self.max_local_steps, layout)
            self.global_step += self.max_local_steps * self.emulator_counts      # global_step += 1 per env per step (paac.py:341)
            stats = net.train_rollout(self.get_lr())
            self._log_update(stats)
            counter += 1
            if pe is not None and time.time() - last_eval >= eval_every:
                pe.eval_once(max_sequence_length=self.rnn_length)
                last_eval = time.time()
            ckpt_every = int(getattr(self, "checkpoint_every", 0) or 0)
            if ckpt_every > 0 and counter % ckpt_every == 0:
                self.save_checkpoint(getattr(self, "checkpoint_path", "checkpoint.npz"))
            if counter % max(1, int(5048 / self.emulator_counts)) == 0:
                curr_time = time.time()
                logging.info("Ran {} steps, at {} steps/s ({} steps/s avg), loss {}"
                             .format(self.global_step,
                                     self.max_local_steps * self.emulator_counts / (curr_time - loop_start_time),
                                     (self.global_step - global_step_start) / (curr_time - start_time), stats["loss"]))
            if max_updates is not None and counter >= max_updates:
                break
        return stats

    def cleanup(self):
        if self.network.net is not None:
            self.network.net.close()
