#!/usr/bin/env python3
"""Golden fixtures of the LEARNER LOOP, captured by running the reference's own code unmodified:

  paac_loop.npz     PAACLearner.__init__/train (fed_gym/agents/paac/paac.py:16-213, actor_learner.py:12-83) on Solow with the
                    real Runners + SolowRunner worker PROCESSES (runners.py:11-54, emulator_runner.py:38-79), and
                    GridPAACLearner.train (paac.py:216-419) on Swarm with the real GridRunners + SwarmRunner processes
                    (emulator_runner.py:120-151): rows R2, R3, R4, R6, quirk Q4 and the flattened feed order of SURVEY 8(a).
  swarm_runner.npz  SwarmRunner._run (emulator_runner.py:120-151) alone, rnn_length 1 and 2: STATE / HISTORY / POSITIONS /
                    REWARD / DONE slots over a TimeLimit-shortened episode incl. the reset observation (row S9).

Run in the build container only:   python tests/golden/gen_golden_learner.py

What is NOT the reference here, and why (TensorFlow 1.4.1 and gym 0.9.4 are absent from the image):
  * `tensorflow` is the stand-in module built below.  It does no arithmetic of the path: the network's outputs are CANNED
    float32 arrays (mu, sigma, vs drawn from a private RandomState by FakeSession.run), the optimizer/saver/placeholder
    objects are inert keys, tf.Summary / FileWriter record what the loop hands them, tf.train.Coordinator never stops.
    The one third-party function with behaviour the loop depends on is keras' pad_sequences (paac.py:88-90, 257-259,
    emulator_runner.py:142), restated from Keras 2.0.8 (`pad_sequences` below; parity unpinned at that call, like TimeLimit).
  * `paac.SolowPolicyMonitor` / `paac.SwarmPolicyMonitor` (module globals of paac.py) are replaced by an inert class: the
    eval thread builds a second TF graph and shares the global numpy generator with the loop; it is SURVEY 8(f)1, pinned
    separately (swarm_reset/swarm_traj fixtures).
  * the loop's local arrays (rewards, values, y_batch, adv_batch, total_rewards, ...) are read from train()'s live frame
    (sys._getframe) by the fake session at the moment train() calls session.run -- nothing of train() is re-typed here.
Everything else -- both train() loops, ActorLearner.__init__, rescale_reward, get_lr, Runners/GridRunners, the worker
processes, the envs and state processors -- is the code under /root/reference, imported, not copied.
"""
import logging
import os
import shutil
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True      # /root/reference is read-only by contract: no __pycache__ next to its sources

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402


# ------------------------------------------------------------------------------------------- tensorflow stand-in
def pad_sequences(sequences, maxlen=None, dtype='int32', padding='pre', truncating='pre', value=0.):
    """Keras 2.0.8 keras/preprocessing/sequence.py:pad_sequences (what tf.keras of TF 1.4.1 exposes), restated:
    default dtype int32 (!), samples keep their trailing shape, truncation keeps the LAST maxlen entries by default."""
    lengths = [len(s) for s in sequences]
    if maxlen is None:
        maxlen = np.max(lengths)
    sample_shape = tuple()
    for s in sequences:
        if len(s) > 0:
            sample_shape = np.asarray(s).shape[1:]
            break
    x = (np.ones((len(sequences), maxlen) + sample_shape) * value).astype(dtype)
    for idx, s in enumerate(sequences):
        if len(s) == 0:
            continue
        trunc = s[-maxlen:] if truncating == 'pre' else s[:maxlen]
        trunc = np.asarray(trunc, dtype=dtype)
        if padding == 'post':
            x[idx, :len(trunc)] = trunc
        else:
            x[idx, -len(trunc):] = trunc
    return x


class Key(object):
    """An inert graph node: placeholders, outputs and ops are only ever used as dict keys / fetch names."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<%s>" % self.name


class _SummaryValues(list):
    def add(self, simple_value=None, tag=None):
        self.append((tag, simple_value))


class Summary(object):
    def __init__(self):
        self.value = _SummaryValues()


class FileWriter(object):
    """tf.summary.FileWriter: records (tag, value, global_step) and, for rl/reward points, the loop's own counters."""
    instances = []

    def __init__(self, logdir):
        self.logdir = logdir
        self.records = []
        self.merged = []        # (what, global_step) of the per-update add_summary(summaries, global_step) calls
        FileWriter.instances.append(self)

    def get_logdir(self):
        return self.logdir

    def add_summary(self, summary, global_step):
        if isinstance(summary, Summary):
            f = sys._getframe(1)
            loc = f.f_locals
            for tag, v in summary.value:
                self.records.append((tag, float(v), int(global_step), int(loc.get("e_idx", -1)), int(loc.get("t", -1)),
                                     type(v).__name__))
        else:
            self.merged.append((str(summary), int(global_step)))

    def flush(self):
        pass


class Coordinator(object):
    def should_stop(self):
        return False

    def request_stop(self):
        pass

    def join(self, threads=None):
        for t in threads or []:
            t.join()


class AdamOptimizer(object):
    def __init__(self, lr, name=None):
        self.lr = lr

    def compute_gradients(self, loss):
        return [(Key("grad0"), Key("var0"))]

    def apply_gradients(self, grads_and_vars, global_step=None):
        return Key("train_step")


def make_tensorflow(session_factory):
    tf = _ref_stubs._Anything("tensorflow")
    tf.train = types.SimpleNamespace(Coordinator=Coordinator, AdamOptimizer=AdamOptimizer, Saver=lambda *a, **k: Key("saver"))
    tf.summary = types.SimpleNamespace(FileWriter=FileWriter, merge_all=lambda: Key("summaries_op"),
                                       scalar=lambda *a, **k: None)
    tf.Summary = Summary
    tf.placeholder = lambda *a, **k: Key("learning_rate")
    tf.concat = lambda parts, axis=0: Key("concat")
    tf.reshape = lambda t, shape: t
    tf.identity = lambda t, name=None: t
    tf.global_norm = lambda ts, name=None: Key("global_norm")
    tf.clip_by_global_norm = lambda ts, clip: (list(ts), Key("global_norm"))
    tf.ConfigProto = lambda **k: types.SimpleNamespace(gpu_options=types.SimpleNamespace(allow_growth=False))
    tf.Session = lambda config=None: session_factory()
    tf.global_variables = lambda: []
    keras = types.SimpleNamespace(preprocessing=types.SimpleNamespace(sequence=types.SimpleNamespace(pad_sequences=pad_sequences)))
    tf.keras = keras
    return tf


# ------------------------------------------------------------------------------------------- fake network / session
class FakeNetwork(object):
    def __init__(self, scale, height=84, channels=3):
        for k in ("mu", "sigma", "vs", "states", "history", "critic_target", "actions", "advantages", "global_step_tensor", "loss"):
            setattr(self, k, Key(k))
        self.scale, self.height, self.channels = scale, height, channels
        self.conf = {}

    def init(self, checkpoint_folder, saver, session):
        return 0

    def predict(self, states, session):          # policy_v_network.py:69-80
        mu, sigma, vs = session.run([self.mu, self.sigma, self.vs], feed_dict={self.states: states})
        return {"mu": mu, "sigma": sigma, "vs": vs}


_TRAIN_LOCALS = ("rewards", "values", "episodes_over_masks", "y_batch", "adv_batch", "actions", "next_state_value",
                 "total_rewards", "emulator_steps", "total_episode_rewards", "lr", "counter")


def _train_frame():
    f = sys._getframe(2)
    while f is not None and f.f_code.co_name != "train":
        f = f.f_back
    return f


class FakeSession(object):
    """session.run with canned float32 outputs.  Records, per call: the arrays fed, the runners' shared variables as the loop
    sees them at that moment, and at every train step the loop's own local arrays."""
    current = None

    def __init__(self):
        self.rng = np.random.RandomState(4242)
        self.calls, self.steps, self.boots, self.updates = 0, [], [], []
        self.net = None
        self.learner = None
        self.num_actions = 1
        self.sparse_states = False
        FakeSession.current = self

    def _shared(self):
        r = self.learner.runners
        return None if r is None else [np.array(v) for v in r.get_shared_variables()]

    def _canned(self, n):
        a = self.num_actions
        mu = (self.rng.normal(size=(n, a)) * 0.7).astype(np.float32)
        sigma = (0.05 + self.rng.rand(n, a)).astype(np.float32)
        vs = (self.rng.normal(size=(n,)) * 3.0).astype(np.float32)
        return mu, sigma, vs

    def run(self, fetches, feed_dict=None):
        net = self.net
        names = [k.name for k in fetches] if isinstance(fetches, (list, tuple)) else fetches.name
        feed = {k.name: np.array(v) for k, v in (feed_dict or {}).items()}
        if names == ["mu", "sigma", "vs"]:
            n = feed["states"].shape[0]
            mu, sigma, vs = self._canned(n)
            shared = self._shared()
            rec = {"mu": mu, "sigma": sigma, "vs": vs, "shared": shared}
            self.steps.append(rec)
            return mu, sigma, vs
        if names == "vs":
            n = feed["states"].shape[0]
            vs = self._canned(n)[2]
            self.boots.append({"vs": vs, "shared": self._shared()})
            return vs
        if names == ["train_step", "summaries_op", "global_step_tensor"]:
            loc = _train_frame().f_locals
            snap = {k: np.array(loc[k]) for k in _TRAIN_LOCALS if k in loc}
            snap["global_step"] = int(loc["self"].global_step)
            snap["feed"] = feed
            self.updates.append(snap)
            return None, "summaries", len(self.updates)
        raise AssertionError("unexpected fetch %r" % (names,))

    def close(self):
        pass


class InertMonitor(object):
    def __init__(self, **kw):
        pass

    def continuous_eval(self, *a, **k):
        return


# ------------------------------------------------------------------------------------------- install + import the reference
sys.modules["tensorflow"] = make_tensorflow(FakeSession)
_ref_stubs.install()

import gym  # noqa: E402
import fed_gym  # noqa: E402,F401
from gym.envs.registration import register  # noqa: E402
from fed_gym.agents.paac import paac, emulator_runner  # noqa: E402
from fed_gym.agents.state_processors import SolowStateProcessor, SwarmStateProcessor  # noqa: E402

from fed_gym.envs import fed_env  # noqa: E402

fed_env.register_solow_env(1, 1)      # as scripts/train_paac_solow.py does before building the learner
paac.SolowPolicyMonitor = InertMonitor
paac.SwarmPolicyMonitor = InertMonitor

# the (1,)-row action adapter of _ref_adapters.py around the unmodified SolowEnv (numpy-2 ragged-array error, see there)
register(id="Solow-golden-cap6-v0", entry_point="_ref_adapters:SolowEnvRowAction", max_episode_steps=6, kwargs=dict(p=1, q=1, seed=1692))
register(id="Swarm-golden-cap4-v0", entry_point="fed_gym.envs:SwarmEnv", max_episode_steps=4, kwargs=dict(seed=192))


class _Creator(object):
    def __init__(self, env_id):
        self.env_id = env_id

    def create_environment(self):
        return gym.envs.make(self.env_id)


def _args(tmp, **kw):
    d = dict(emulator_workers=2, rnn_length=5, max_local_steps=5, num_actions=1, initial_lr=1e-3, lr_annealing_steps=40000,
             emulator_counts=6, device='/cpu:0', debugging_folder=tmp, max_global_steps=0, gamma=0.99,
             clip_norm_type='global', clip_norm=40.0)
    d.update(kw)
    return types.SimpleNamespace(**d)


class _LogCapture(logging.Handler):
    def __init__(self):
        super().__init__(level=logging.INFO)
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def _run_learner(cls, env_id, emulator_class, state_processor, scale, n_updates, **kw):
    tmp = tempfile.mkdtemp(prefix="grl_golden_")
    try:
        a = _args(tmp, **kw)
        a.max_global_steps = n_updates * a.max_local_steps * a.emulator_counts
        net = FakeNetwork(scale)
        FileWriter.instances.clear()
        np.random.seed(20260)                 # the loop's action noise (paac.py:36, 418) comes from the global generator
        learner = cls(lambda: net, _Creator(env_id), a, emulator_class, state_processor)
        sess = learner.session
        sess.net, sess.learner, sess.num_actions = net, learner, a.num_actions
        cap = _LogCapture()
        root = logging.getLogger()
        old_level = root.level
        root.addHandler(cap); root.setLevel(logging.INFO)
        try:
            learner.train()
        finally:
            root.removeHandler(cap); root.setLevel(old_level)
        return learner, sess, FileWriter.instances[0], cap.lines, a
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _sparse(a):
    idx = np.argwhere(a != 0).astype(np.int32)
    return idx, a[tuple(idx.T)]


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %7d bytes  %d keys" % (name + ".npz", os.path.getsize(path), len(arrays)))


# ------------------------------------------------------------------------------------------- paac_loop.npz
def paac_loop_fixture():
    out = {}
    # ---- flat learner: Solow, masked + clipped returns (paac.py:119-207).  5048 / E = 4 -> log line at update 4
    E, T, n_upd = 1262, 3, 4
    learner, sess, writer, log, a = _run_learner(paac.PAACLearner, "Solow-golden-cap6-v0", emulator_runner.SolowRunner,
                                                 SolowStateProcessor(), scale=10.0, n_updates=n_upd, emulator_counts=E,
                                                 max_local_steps=T, rnn_length=5, num_actions=1)
    assert len(sess.updates) == n_upd and len(sess.steps) == n_upd * T and len(sess.boots) == n_upd
    KEEP = 8        # envs whose per-step slots are kept in full (all envs share one seeded tape; actions differ per env)
    out["flat_E"], out["flat_T"], out["flat_updates"] = np.array(E), np.array(T), np.array(n_upd)
    out["flat_gamma"], out["flat_scale"] = np.array(a.gamma), np.array(10.0)
    out["flat_lr0"], out["flat_anneal"], out["flat_cap"] = np.array(a.initial_lr), np.array(a.lr_annealing_steps), np.array(6)
    out["flat_mu"] = np.array([s["mu"] for s in sess.steps])
    out["flat_sigma"] = np.array([s["sigma"] for s in sess.steps])
    out["flat_vs"] = np.array([s["vs"] for s in sess.steps])
    out["flat_boot"] = np.array([b["vs"] for b in sess.boots])
    # shared variables as the loop saw them when it asked for the next action (= after the previous step) / for the bootstrap
    for j, nm in enumerate(("states", "hist", "rew", "done", "act")):
        out["flat_shared_" + nm] = np.array([s["shared"][j][:KEEP] for s in sess.steps])
        out["flat_bootshared_" + nm] = np.array([b["shared"][j][:KEEP] for b in sess.boots])
    # all envs: the reward / done / action slots AFTER each of the n_upd*T steps (what the loop reads at paac.py:140-142)
    post = [s["shared"] for s in sess.steps[1:]] + [sess.boots[-1]["shared"]]
    out["flat_post_rew"] = np.array([p[2] for p in post])
    out["flat_post_done"] = np.array([p[3] for p in post])
    out["flat_post_act"] = np.array([p[4] for p in post])
    out["flat_post_states"] = np.array([p[0] for p in post])
    for k in ("rewards", "values", "episodes_over_masks", "y_batch", "adv_batch", "actions"):
        out["flat_" + k] = np.array([u[k] for u in sess.updates])
    out["flat_lr"] = np.array([u["lr"] for u in sess.updates])
    out["flat_global_step"] = np.array([u["global_step"] for u in sess.updates])
    out["flat_total_rewards_final"] = np.array(sess.updates[-1]["total_rewards"], np.float64)
    out["flat_running_total"] = np.array(sess.updates[-1]["total_episode_rewards"], np.float64)
    out["flat_running_steps"] = np.array(sess.updates[-1]["emulator_steps"], np.int64)
    u0 = sess.updates[0]["feed"]
    for k in ("states", "history", "critic_target", "actions", "advantages"):
        out["flat_feed0_" + k] = u0[k]
    out["flat_feed_lr"] = np.array([u["feed"]["learning_rate"] for u in sess.updates])
    out["flat_feed_critic_target"] = np.array([u["feed"]["critic_target"] for u in sess.updates])
    out["flat_feed_advantages"] = np.array([u["feed"]["advantages"] for u in sess.updates])
    recs = [r for r in writer.records if r[0] == "rl/reward"]
    out["flat_rl_reward"] = np.array([r[1] for r in recs])
    out["flat_rl_step"] = np.array([r[2] for r in recs], np.int64)
    out["flat_rl_env"] = np.array([r[3] for r in recs], np.int64)
    out["flat_rl_t"] = np.array([r[4] for r in recs], np.int64)
    out["flat_rl_value_type"] = np.array(sorted({r[5] for r in recs}))
    out["flat_merged_summary_steps"] = np.array([m[1] for m in writer.merged], np.int64)   # global_step_tensor after each update
    line = [l for l in log if l.startswith("Ran ")]
    assert len(line) == 1, log
    out["flat_log_steps"] = np.array(int(line[0].split()[1]))
    out["flat_log_last_ten"] = np.array(float(line[0].rsplit("avg ", 1)[1]))
    # the seeded env's episode (every reset replays it): z0 and shock tape, for replay on the device engine
    env = gym.envs.make("Solow-golden-cap6-v0")
    env.reset()
    out["flat_tape"] = np.array(env.unwrapped.es)
    out["flat_z0"], out["flat_e0"], out["flat_k0"] = env.unwrapped.z.copy(), env.unwrapped.e.copy(), np.array(env.unwrapped.k)

    # ---- grid learner: Swarm, unmasked unclipped returns, reward columns e_idx < E only (quirk Q4) (paac.py:302-401)
    E, T, n_upd = 4, 3, 2
    learner, sess, writer, log, a = _run_learner(paac.GridPAACLearner, "Swarm-golden-cap4-v0", emulator_runner.SwarmRunner,
                                                 SwarmStateProcessor(grid_size=84), scale=1000.0, n_updates=n_upd,
                                                 emulator_counts=E, max_local_steps=T, rnn_length=1, num_actions=2,
                                                 emulator_workers=2)
    assert len(sess.updates) == n_upd and len(sess.steps) == n_upd * T
    out["grid_E"], out["grid_T"], out["grid_updates"] = np.array(E), np.array(T), np.array(n_upd)
    out["grid_gamma"], out["grid_scale"], out["grid_cap"] = np.array(a.gamma), np.array(1000.0), np.array(4)
    out["grid_mu"] = np.array([s["mu"] for s in sess.steps])
    out["grid_sigma"] = np.array([s["sigma"] for s in sess.steps])
    out["grid_vs"] = np.array([s["vs"] for s in sess.steps])
    out["grid_boot"] = np.array([b["vs"] for b in sess.boots])
    st = np.array([s["shared"][0] for s in sess.steps] + [sess.boots[-1]["shared"][0]])     # (n_upd*T + 1, E,10,84,84,3)
    out["grid_shared_states_idx"], out["grid_shared_states_val"] = _sparse(st)
    out["grid_shared_states_shape"] = np.array(st.shape)
    out["grid_shared_hist_shape"] = np.array(sess.steps[0]["shared"][1].shape)
    # rnn_length 1: the history slot (float32 shared array, paac.py:257-262) always holds the current state
    out["grid_shared_hist_equals_states_f32"] = np.array(all(
        np.array_equal(s["shared"][1].reshape(s["shared"][0].shape), s["shared"][0].astype(np.float32)) for s in sess.steps))
    for j, nm in ((2, "pos"), (3, "rew"), (4, "done"), (5, "act")):
        out["grid_shared_" + nm] = np.array([s["shared"][j] for s in sess.steps] + [sess.boots[-1]["shared"][j]])
    out["grid_shared_dtypes"] = np.array([str(v.dtype) for v in sess.steps[0]["shared"]])
    for k in ("rewards", "values", "y_batch", "adv_batch", "actions"):
        out["grid_" + k] = np.array([u[k] for u in sess.updates])
    out["grid_lr"] = np.array([u["lr"] for u in sess.updates])
    out["grid_global_step"] = np.array([u["global_step"] for u in sess.updates])
    out["grid_total_rewards_final"] = np.array(sess.updates[-1]["total_rewards"], np.float64)
    out["grid_running_total"] = np.array(sess.updates[-1]["total_episode_rewards"], np.float64)
    out["grid_running_steps"] = np.array(sess.updates[-1]["emulator_steps"], np.int64)
    f0 = sess.updates[0]["feed"]
    fs = f0["states"]                                   # (T*B, 84, 84, 3) time-major, env-major inside a step
    out["grid_feed0_states_idx"], out["grid_feed0_states_val"] = _sparse(fs)
    out["grid_feed0_states_shape"] = np.array(fs.shape)
    for k in ("critic_target", "actions", "advantages"):
        out["grid_feed_" + k] = np.array([u["feed"][k] for u in sess.updates])
    out["grid_feed_keys"] = np.array(sorted(f0.keys()))
    recs = [r for r in writer.records if r[0] == "rl/reward"]
    out["grid_rl_reward"] = np.array([r[1] for r in recs])
    out["grid_rl_step"] = np.array([r[2] for r in recs], np.int64)
    out["grid_rl_env"] = np.array([r[3] for r in recs], np.int64)
    out["grid_rl_t"] = np.array([r[4] for r in recs], np.int64)
    save("paac_loop", **out)


# ------------------------------------------------------------------------------------------- swarm_runner.npz
def swarm_runner_fixture():
    """SwarmRunner._run driven directly (in this process), one instruction per call, for rnn_length 1 and 2."""

    class Q(object):
        def __init__(self, n): self.n = n
        def get(self):
            self.n -= 1
            return True if self.n >= 0 else None
        def put(self, _): pass

    out = {}
    sp = SwarmStateProcessor(grid_size=84)
    for rnn in (1, 2):
        E, steps = 2, 7
        emulators = [gym.envs.make("Swarm-golden-cap4-v0") for _ in range(E)]
        init, pos = [], []
        for e in emulators:
            s = sp.process_state(e.reset())
            init.append(emulator_runner.SwarmRunner.get_local_states(s, sp.positions)); pos.append(sp.positions.copy())
        init = np.array(init)
        tsm = pad_sequences(np.expand_dims(init.reshape(-1, 84, 84, 3), 1), dtype='float32', padding='post', maxlen=rnn)
        tsm = np.reshape(tsm, (E, 10, rnn, 84, 84, 3))                      # paac.py:255-262
        variables = [init.copy(), tsm.astype(np.float64), np.array(pos).astype(np.uint32),
                     np.zeros((E, 10), np.float32), np.zeros((E, 10), np.float32), np.zeros((E, 10, 2), np.float32)]
        runner = emulator_runner.SwarmRunner(0, emulators, variables, Q(0), Q(0), 84)
        rng = np.random.RandomState(31 + rnn)
        raw = rng.normal(size=(steps, E, 10, 2))
        rec = {k: [] for k in ("pos", "rew", "done", "act", "x", "xa")}
        states, hists = [], []
        for t in range(steps):
            act = emulator_runner.SwarmRunner.transform_actions_for_env(raw[t].reshape(-1, 2).copy()).reshape(E, 10, 2)
            variables[-1][:] = act.astype(np.float32)
            runner.queue = Q(1)
            runner._run()
            states.append(variables[0].copy()); hists.append(variables[1].copy())
            rec["pos"].append(variables[2].copy()); rec["rew"].append(variables[3].copy())
            rec["done"].append(variables[4].copy()); rec["act"].append(variables[5].copy())
            rec["x"].append(np.array([e.unwrapped.states[0] for e in emulators]))      # raw env state behind the slots
            rec["xa"].append(np.array([e.unwrapped.states[1] for e in emulators]))
        k = "r%d_" % rnn
        out[k + "raw_actions"] = raw
        out[k + "init_states_idx"], out[k + "init_states_val"] = _sparse(init)
        out[k + "init_pos"] = np.array(pos)
        st = np.array(states)
        out[k + "states_idx"], out[k + "states_val"] = _sparse(st)
        out[k + "states_shape"] = np.array(st.shape)
        hs = np.array(hists)
        out[k + "hist_idx"], out[k + "hist_val"] = _sparse(hs)
        out[k + "hist_shape"] = np.array(hs.shape)
        for nm in rec:
            out[k + nm] = np.array(rec[nm])
    save("swarm_runner", **out)


if __name__ == "__main__":
    paac_loop_fixture()
    swarm_runner_fixture()
