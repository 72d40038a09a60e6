"""Eval monitors (reference fed_gym/agents/paac/policy_monitor.py:15-208) on the device engine.

Same classes, constructor arguments and return values: `eval_once` copies the learner's parameters into a local
estimator, plays ONE episode of the eval env (`Swarm-eval-v0`, seed 192 / a Solow eval env) with actions
a = mu + sigma * N(0,1) drawn from numpy's global generator as the reference does, writes the `eval/*` scalars and, for
Swarm, keeps the best episode's actions in `swarm-eval.json` ({'score', 'actions'}: the file
scripts/make_swarm_gif.py:62-82 replays).  What has no counterpart: the gym `Monitor` video wrapper, TF sessions and
summaries (scalars go to a JSON-lines file), and the Saver (use ActorLearner.save_checkpoint).  The eval env owns
its own 1-env engine handle, so a monitor may run beside the learner."""
import json
import logging
import os
import time

import numpy as np

from .emulator_runner import SolowRunner, SwarmRunner
from .policy_v_network import ConvSingleAgentPolicyNetwork, FlatPolicyVNetwork


class ScalarWriter(object):
    """Stand-in for tf.summary.FileWriter: every scalar goes to <logdir>/scalars.jsonl (one JSON object per line) and to a
    TensorBoard event file (events.out.tfevents.*, written by goldsrl.utils_tfevents without TensorFlow)."""

    def __init__(self, logdir, tfevents=True):
        self.logdir = os.path.abspath(logdir)
        os.makedirs(self.logdir, exist_ok=True)
        self._f = open(os.path.join(self.logdir, "scalars.jsonl"), "a")
        self._ev = None
        if tfevents:
            from ...utils_tfevents import EventFileWriter
            self._ev = EventFileWriter(self.logdir)

    def get_logdir(self):
        return self.logdir

    def add_scalar(self, tag, value, step):
        now = time.time()
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step), "time": now}) + "\n")
        if self._ev is not None:
            self._ev.add_scalar(tag, value, step, wall_time=now)

    def flush(self):
        self._f.flush()
        if self._ev is not None:
            self._ev.flush()

    def close(self):
        self._f.close()
        if self._ev is not None:
            self._ev.close()


class PolicyMonitor(object):
    """policy_monitor.py:15-71.  `global_policy_net` is the learner's estimator object (bound to its engine);
    `learner` (optional) supplies global_step for the scalars."""

    def __init__(self, env, global_policy_net, state_processor, summary_writer, saver=None, network_conf=None, learner=None):
        self.env = env
        self.state_processor = state_processor
        self.global_policy_net = global_policy_net
        self.summary_writer = summary_writer
        self.saver = saver
        self.learner = learner
        self.best_score = -np.inf
        logdir = summary_writer.get_logdir() if summary_writer is not None else "."
        self.checkpoint_path = os.path.abspath(os.path.join(logdir, "../checkpoints/model"))
        self.actions_path = os.path.join(os.getcwd(), 'swarm-eval.json')
        self.policy_net = self._create_policy_estimator(network_conf)      # the "policy_eval" copy
        self._bind()

    def _bind(self):
        raise NotImplementedError

    def copy_params(self):
        """copy_params_op: global -> policy_eval"""
        self.policy_net.set_flat_params(self.global_policy_net.get_flat_params())
        return int(self.learner.global_step) if self.learner is not None else 0

    def get_action_from_policy(self, processed_state, history, positions, sess=None):
        predictions = self.policy_net.predict(processed_state, history)
        mu, sigma = predictions['mu'], predictions['sigma']
        return mu + sigma * np.random.normal(size=mu.shape)

    @staticmethod
    def _create_policy_estimator(conf):
        raise NotImplementedError

    def eval_once(self, sess=None, max_sequence_length=5):
        raise NotImplementedError

    def _play(self, first_state, choose_action):
        """One episode: `choose_action(state, t)` -> env action until the env reports done.  Returns the per-step rewards
        and the actions taken."""
        rewards, taken = [], []
        state, done = first_state, False
        while not done:
            action = choose_action(state, len(rewards))
            taken.append(action)
            state, reward, done, _ = self.env.step(action)
            rewards.append(reward)
        return rewards, taken

    def _summaries(self, global_step, total_reward, episode_length, rewards):
        if self.summary_writer is not None:
            self.summary_writer.add_scalar("eval/total_reward", total_reward, global_step)
            self.summary_writer.add_scalar("eval/episode_length", episode_length, global_step)
            self.summary_writer.flush()
        logging.info("Eval results at step {}: avg_reward {}, std_reward {}, episode_length {}".format(
            global_step, np.mean(rewards), np.std(rewards), episode_length))

    def continuous_eval(self, eval_every, sess=None, coord=None, max_seq_length=5):
        """Evaluates the policy every [eval_every] seconds until coord.should_stop()."""
        while coord is None or not coord.should_stop():
            self.eval_once(sess, max_sequence_length=max_seq_length)
            if coord is None:
                return
            time.sleep(eval_every)


class SolowPolicyMonitor(PolicyMonitor):
    """policy_monitor.py:74-124"""

    def _bind(self):
        self.policy_net.bind(self.env._eng, max_samples=64)

    def get_action_from_policy(self, processed_state, history, positions, sess=None):
        raw_actions = super().get_action_from_policy(processed_state, history, positions, sess)
        return SolowRunner.transform_actions_for_env(raw_actions)

    @staticmethod
    def _create_policy_estimator(conf):
        return FlatPolicyVNetwork(conf)

    def eval_once(self, sess=None, max_sequence_length=5):
        global_step = self.copy_params()
        window_rows = []       # processed states of the episode so far (the reference keeps the last 2*max_sequence_length)

        def choose(state, t):
            processed = self.state_processor.process_state(state)
            window_rows.append(np.array(processed))
            del window_rows[:-2 * max_sequence_length]
            # the estimator takes a fixed (1, rnn_length, 2) window; rows past the episode's start are zero
            # (dynamic_rnn's sequence_length = number of non-zero rows, a3c/estimators.py:11-15)
            window = np.zeros((1, max_sequence_length, len(processed)), np.float32)
            recent = window_rows[-max_sequence_length:]
            window[0, :len(recent)] = np.array(recent)
            return self.get_action_from_policy(np.array([processed]), window, None, sess)

        rewards, _ = self._play(self.env.reset(), choose)
        total_reward, episode_length = float(np.sum(rewards)), len(rewards)
        self._summaries(global_step, total_reward, episode_length, rewards)
        return total_reward, episode_length, rewards


class SwarmPolicyMonitor(PolicyMonitor):
    """policy_monitor.py:127-208"""

    def _bind(self):
        self.policy_net.bind(self.env._eng)

    def get_action_from_policy(self, processed_state, history, positions, sess=None):
        # the device net reads the eval engine's current observation (process_state + get_local_states run on the device)
        predictions = self.policy_net.predict()
        mu, sigma = predictions['mu'], predictions['sigma']
        # policy_monitor.py:132 draws np.random.normal from the GLOBAL stream, which Swarm-eval-v0's reset has just reseeded
        # (multiagent.py:47-48): the seeded facade exposes that stream as env.np_random
        rng = getattr(self.env, "np_random", None) or np.random
        raw_actions = mu + sigma * rng.normal(size=mu.shape)
        return SwarmRunner.transform_actions_for_env(raw_actions)

    @staticmethod
    def _create_policy_estimator(conf):
        return ConvSingleAgentPolicyNetwork(conf)

    def _save_actions(self, score, actions):
        with open(self.actions_path, 'w') as f:
            json.dump({'score': score, 'actions': actions}, f)

    def eval_once(self, sess=None, max_sequence_length=5, actions=None):
        """`actions`: a queue.Queue of (10,2) arrays replayed instead of the policy (policy_monitor.py:173-176)."""
        global_step = self.copy_params()

        def choose(state, t):
            if actions:
                return np.asarray(actions.get())
            return self.get_action_from_policy(None, None, None, sess)

        rewards, taken = self._play(self.env.reset(), choose)
        total_reward, episode_length = float(np.sum(rewards)), len(rewards)
        if total_reward > self.best_score:
            self.best_score = total_reward
            self._save_actions(total_reward, [np.asarray(a).tolist() for a in taken])
        self._summaries(global_step, total_reward, episode_length, rewards)
        return total_reward, episode_length, rewards
