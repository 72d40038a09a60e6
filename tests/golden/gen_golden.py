#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the UNMODIFIED reference
(/root/reference, read-only) under the gym/tensorflow stand-ins of _ref_stubs.py.

Run in the build container only:   python tests/golden/gen_golden.py
The GPU box never sees /root/reference; it only sees the .npz files written here.
Fixtures are data (inputs + expected outputs); no reference source text is stored.

Each block names the reference call it captures (file:line under /root/reference).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()

import gym  # noqa: E402  (the stub)
import fed_gym  # noqa: E402,F401  (runs the registrations, fed_gym/__init__.py:3-33)
from fed_gym.envs import fed_env, multiagent  # noqa: E402
from fed_gym.agents.state_processors import SolowStateProcessor, SwarmStateProcessor  # noqa: E402
from fed_gym.agents.a3c import worker as a3c_worker  # noqa: E402
from fed_gym.agents.paac import emulator_runner  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %7d bytes  keys=%s" % (name + ".npz", os.path.getsize(path), sorted(arrays)))


# --------------------------------------------------------------------------- Swarm
def swarm_reset_fixture():
    """SwarmEnv._reset (envs/multiagent.py:46-63): draw order + 10 burn-in steps."""
    out = {}
    for seed in (192, 7):
        np.random.seed(seed)
        x0 = np.random.rand(80, 2)
        xa0 = np.random.rand(10, 2)
        random_actions = np.random.normal(size=(10, 10, 2))
        agent_noise = np.random.normal(size=(138, 10, 2))
        particle_noise = np.random.normal(size=(138, 80, 2))
        env = multiagent.SwarmEnv(seed=seed)
        x, xa = env.reset()
        assert np.array_equal(env.agent_noise, agent_noise)
        assert env.t == 10
        k = "s%d_" % seed
        out[k + "x0"], out[k + "xa0"] = x0, xa0
        out[k + "random_actions"] = random_actions
        out[k + "agent_noise"] = agent_noise[:11]
        out[k + "particle_noise"] = particle_noise[:11]
        out[k + "x"], out[k + "xa"] = x.copy(), xa.copy()
    save("swarm_reset", **out)


def swarm_step_fixture():
    """SwarmEnv._step (envs/multiagent.py:30-44) teacher-forced single steps: each case
    carries its own input state, action and noise row, so errors never compound."""
    rng = np.random.RandomState(20240)
    xs, xas, acts, an, pn, xo, xao, rew, done = ([] for _ in range(9))
    for seed in (192, 7, 1692, 3):
        env = multiagent.SwarmEnv(seed=seed)
        env.reset()
        for step in range(60):
            action = rng.normal(size=(10, 2)) * (0.0 if step % 7 == 3 else 1.0)
            if step % 5 == 0:  # capture (also late steps where most locusts sit on the ground)
                x_in, xa_in = env.states[0].copy(), env.states[1].copy()
                a_row = env.agent_noise[env.t].copy()
                p_row = env.particle_noise[env.t].copy()
            (x, xa), r, d, _ = env.step(action)
            if step % 5 == 0:
                xs.append(x_in); xas.append(xa_in); acts.append(action.copy())
                an.append(a_row); pn.append(p_row)
                xo.append(x.copy()); xao.append(xa.copy()); rew.append(r); done.append(d)
    save("swarm_step", x=np.array(xs), xa=np.array(xas), action=np.array(acts),
         agent_noise=np.array(an), particle_noise=np.array(pn),
         x_out=np.array(xo), xa_out=np.array(xao), reward=np.array(rew), done=np.array(done))


def swarm_traj_fixture():
    """Swarm-eval-v0 (seed 192) through the TimeLimit(128) wrapper, 130 wrapped steps
    with the worker's auto-reset rule (paac/emulator_runner.py:126-132)."""
    env = gym.envs.make("Swarm-eval-v0")
    env.reset()
    inner = env.unwrapped
    rng = np.random.RandomState(99)
    actions = rng.normal(size=(130, 10, 2))
    for i in range(len(actions)):  # the runner's norm clip, as the learner applies it
        actions[i] = emulator_runner.SwarmRunner.transform_actions_for_env(actions[i])
    x0, xa0 = inner.states[0].copy(), inner.states[1].copy()
    a_row, p_row = inner.agent_noise[10].copy(), inner.particle_noise[10].copy()
    xs, xas, rewards, dones = [], [], [], []
    for i in range(len(actions)):
        (x, xa), r, d, _ = env.step(actions[i])
        rewards.append(r); dones.append(d)
        if d:
            x, xa = env.reset()
        if i % 16 == 15 or d or i == len(actions) - 1:
            xs.append(x.copy()); xas.append(xa.copy())
    save("swarm_traj", x0=x0, xa0=xa0, agent_noise_row=a_row, particle_noise_row=p_row,
         actions=actions, rewards=np.array(rewards), dones=np.array(dones),
         x_snap=np.array(xs), xa_snap=np.array(xas),
         snap_steps=np.array([i for i in range(130) if i % 16 == 15 or dones[i] or i == 129]))


def swarm_obs_fixture():
    """SwarmStateProcessor.process_state (agents/state_processors.py:29-42) and
    SwarmRunner.get_local_states (paac/emulator_runner.py:98-111)."""
    sp = SwarmStateProcessor(grid_size=84)
    states = []
    env = multiagent.SwarmEnv(seed=192)
    states.append([a.copy() for a in env.reset()])
    rng = np.random.RandomState(5)
    for step in range(100):
        s, _, _, _ = env.step(rng.normal(size=(10, 2)))
        if step in (0, 20, 60, 99):
            states.append([a.copy() for a in s])
    # agent far outside the box on both sides + above the box (Q3: no density, still a one-hot)
    s = [a.copy() for a in states[2]]
    s[1][0] = [s[0][:, 0].mean() + 40.0, 7.5]
    s[1][1] = [s[0][:, 0].mean() - 40.0, 0.0]
    s[0][5, 1] = 6.0            # exactly on the top edge -> last bin
    s[0][6, 1] = 6.0000001      # above -> dropped
    s[0][7, 1] = 3 * (6.0 / 84)  # exactly on an interior y edge
    states.append(s)
    # a locust exactly on an interior x edge: iterate to the fixed point of "x = edge_k(mean(x))"
    s = [a.copy() for a in states[1]]
    for _ in range(50):
        mean_x = np.mean(np.vstack(s), axis=0)[0]
        edges = np.linspace(mean_x - 1.5, mean_x + 1.5, 85)
        if s[0][11, 0] == edges[40] and s[1][3, 0] == edges[17]:
            break
        s[0][11, 0] = edges[40]
        s[1][3, 0] = edges[17]
    on_edge = bool(s[0][11, 0] == edges[40] and s[1][3, 0] == edges[17])
    states.append(s)
    # synthetic wide cloud
    r2 = np.random.RandomState(11)
    states.append([r2.normal(size=(80, 2)) * [1.2, 2.5] + [3.0, 2.0], r2.normal(size=(10, 2)) * [1.5, 3.0] + [3.0, 2.0]])

    grids, poss, xs, xas = [], [], [], []
    for s in states:
        g = sp.process_state([s[0], s[1]])
        grids.append(g.copy()); poss.append(sp.positions.copy())
        xs.append(s[0]); xas.append(s[1])
    local = emulator_runner.SwarmRunner.get_local_states(grids[0], poss[0])
    onehot_idx = np.array([np.argwhere(l[:, :, 2] == 1.0)[0] for l in local])
    assert all(np.array_equal(l[:, :, :2], grids[0]) for l in local)
    assert all(l[:, :, 2].sum() == 1.0 for l in local)
    save("swarm_obs", x=np.array(xs), xa=np.array(xas), grid=np.array(grids),
         positions=np.array(poss), local0_onehot_idx=onehot_idx, on_edge_case_exact=np.array(on_edge))


def swarm_action_fixture():
    """SwarmRunner.transform_actions_for_env (paac/emulator_runner.py:113-118)."""
    rng = np.random.RandomState(3)
    a = rng.normal(size=(64, 2))
    a[0] = [0.6, 0.8]       # norm exactly 1 (>= branch)
    a[1] = [0.3, 0.4]
    a[2] = [3.0, -4.0]
    a[3] = [0.0, 0.0]
    a[4] = [1.0, 0.0]
    out64 = emulator_runner.SwarmRunner.transform_actions_for_env(a.copy())
    a32 = a.astype(np.float32)
    out32 = emulator_runner.SwarmRunner.transform_actions_for_env(a32.copy())
    save("swarm_action", a64=a, out64=out64, a32=a32, out32=out32)


# --------------------------------------------------------------------------- Solow
def solow_fixture():
    """SolowEnv._reset/_step (envs/fed_env.py:201-250), SolowSSEnv (:253-265),
    SolowStateProcessor (agents/state_processors.py:11-12,69-71)."""
    out = {}
    for (p, q) in ((1, 1), (3, 2)):
        env = fed_env.SolowEnv(p=p, q=q, seed=1692)
        obs0 = env.reset()
        tape = np.array(env.es)          # consumed from the END (es.pop(): fed_env.py:206)
        z0, e0 = env.z.copy(), env.e.copy()
        rng = np.random.RandomState(4)
        s_seq = np.concatenate([np.full(8, 0.1), rng.rand(24), [0.0, 1e-4, 0.999, 1.0]])
        obs, rew, zs, es = [], [], [], []
        for s in s_seq:
            o, r, d, _ = env.step(float(s))
            assert d is False
            obs.append(o); rew.append(r); zs.append(env.z.copy()); es.append(env.e.copy())
        k = "p%dq%d_" % (p, q)
        out[k + "rho_z"], out[k + "rho_e"] = env.rho_z, env.rho_e
        out[k + "obs0"], out[k + "z0"], out[k + "e0"] = obs0, z0, e0
        out[k + "tape_tail"] = tape[-64:]
        out[k + "s"] = s_seq
        out[k + "obs"], out[k + "reward"] = np.array(obs), np.array(rew)
        out[k + "z"], out[k + "e"] = np.array(zs), np.array(es)
    # analytical steady state (tests/env_tests.py:145-155)
    ss = fed_env.SolowSSEnv(sigma=0.0, T=10000)
    try:
        ss.reset()
    except ValueError:
        # fed_env.py:265 builds a ragged np.array([k, z]) which numpy>=1.24 rejects; every
        # field of the env is already assigned by then (fed_env.py:258-263), so carry on.
        pass
    out["ss_k0"] = np.array(ss.k)
    for _ in range(10000):
        o, c, d, _ = ss.step(0.1)
    out["ss_capital_10000"] = np.array(o[0])
    out["ss_k_closed_form"] = np.array((0.1 / ss.delta) ** (1 / (1 - ss.alpha)))
    out["ss_last_reward"] = np.array(c)
    out["k_ss_033"] = np.array(fed_env.SolowEnv()._k_ss(0.33))
    sp = SolowStateProcessor()
    out["proc_in"] = np.array([65.6357, -0.123])
    out["proc_out"] = sp.process_state(out["proc_in"])
    save("solow", **out)


def solow_runner_fixture():
    """EmulatorRunner._run body for SolowRunner (paac/emulator_runner.py:38-79): sigmoid'ed
    actions come from the learner; worker does step / auto-reset / process_state / 5-deep
    history window zero-padded at the end.  TimeLimit shortened to 6 so a reset happens."""
    from gym.envs.registration import register
    register(id="Solow-golden-short-v0", entry_point="fed_gym.envs:SolowEnv",
             max_episode_steps=6, kwargs=dict(p=1, q=1, seed=1692))
    E, rnn, steps = 3, 5, 15
    emulators = [gym.envs.make("Solow-golden-short-v0") for _ in range(E)]
    sp = SolowStateProcessor()
    init = [sp.process_state(e.reset()) for e in emulators]
    hist0 = np.zeros((E, rnn, 2)); hist0[:, 0, :] = np.array(init)
    variables = [np.array(init), hist0.copy(), np.zeros(E, np.float32), np.zeros(E, np.float32),
                 np.zeros((E,), np.float32)]   # (E,) not (E,1): under numpy>=1.24 a (1,)-shaped
    # action makes fed_env.py:229 build a ragged array (ordinary ValueError); scalars are fine

    class Q(object):
        def __init__(self, n): self.n = n
        def get(self):
            self.n -= 1
            return True if self.n >= 0 else None
        def put(self, _): pass

    rng = np.random.RandomState(8)
    raw = rng.normal(size=(steps, E, 1))
    rec = {k: [] for k in ("states", "hist", "rew", "done")}
    runner = emulator_runner.SolowRunner(0, emulators, variables, Q(0), Q(0))
    # inject each env's full shock tape so the build can replay resets (seeded => same tape every reset)
    tapes = np.array([np.array(e.unwrapped.es) for e in emulators])
    z0 = np.array([e.unwrapped.z.copy() for e in emulators])
    for t in range(steps):
        variables[-1][:] = emulator_runner.SolowRunner.transform_actions_for_env(raw[t]).astype(np.float32)[:, 0]
        runner.queue = Q(1)
        runner._run()
        rec["states"].append(variables[0].copy()); rec["hist"].append(variables[1].copy())
        rec["rew"].append(variables[2].copy()); rec["done"].append(variables[3].copy())
    save("solow_runner", raw_actions=raw, init_states=np.array(init), tapes=tapes, z0=z0,
         states=np.array(rec["states"]), hist=np.array(rec["hist"]),
         rew=np.array(rec["rew"]), done=np.array(rec["done"]))


# --------------------------------------------------------------------------- TradeAR1
def trade_fixture():
    """TradeAR1Env._reset/_step (envs/fed_env.py:300-330); TradeWorker.process_state /
    transform_raw_action (agents/a3c/worker.py:420-442)."""
    out = {}
    for n in (2, 16):
        env = fed_env.TradeAR1Env(n_assets=n)
        obs0 = env.reset()
        steps = 40
        np.random.seed(77 + n)
        normals = np.random.normal(size=(steps, n))   # what _price_transition will draw
        np.random.seed(77 + n)
        rng = np.random.RandomState(n)
        acts = np.tanh(rng.normal(size=(steps, n)))
        acts[5] = 0.0
        acts[6] = -1.0
        acts[7] = 1.0
        obs, rew, done = [], [], []
        for t in range(steps):
            o, r, d, _ = env.step(acts[t].copy())
            obs.append(o); rew.append(r); done.append(d)
        k = "n%d_" % n
        out[k + "std_e"] = np.array(env.std_e)
        out[k + "obs0"], out[k + "normals"], out[k + "actions"] = obs0, normals, acts
        out[k + "obs"], out[k + "reward"], out[k + "done"] = np.array(obs), np.array(rew), np.array(done)
    # depletion to done: buy everything then let the price noise (forced strongly negative) sink assets < 1
    env = fed_env.TradeAR1Env(n_assets=2)
    env.reset()
    np.random.seed(5)
    normals = -np.abs(np.random.normal(size=(400, 2))) * 60.0
    obs, rew, done, acts = [], [], [], []
    orig = np.random.normal
    it = iter(normals)
    np.random.normal = lambda size=None: next(it)
    try:
        for t in range(400):
            a = np.array([1.0, 1.0]) if t < 3 else np.array([0.0, 0.0])
            o, r, d, _ = env.step(a.copy())
            obs.append(o); rew.append(r); done.append(d); acts.append(a)
            if d:
                break
    finally:
        np.random.normal = orig
    assert done[-1], "depletion case must reach done"
    out["dep_normals"] = normals[:len(obs)]
    out["dep_actions"], out["dep_obs"] = np.array(acts), np.array(obs)
    out["dep_reward"], out["dep_done"] = np.array(rew), np.array(done)
    # obs/action transforms
    raw = np.array(out["n2_obs"][:8])
    out["proc_in"] = raw
    out["proc_out"] = np.array([a3c_worker.TradeWorker.process_state(None, r) for r in raw])
    x = np.linspace(-4, 4, 33)
    out["tanh_in"], out["tanh_out"] = x, a3c_worker.TradeWorker.transform_raw_action(x)
    save("trade", **out)


# --------------------------------------------------------------------------- rollout math
def returns_fixture():
    """A3C GAE (agents/a3c/worker.py:232-239, 284-300), rescale_reward
    (paac/actor_learner.py:91-97), get_lr (:115-119), sigmoid (a3c/worker.py:17-34)."""
    from fed_gym.agents.paac.actor_learner import ActorLearner
    rng = np.random.RandomState(12)
    T, B, gamma = 20, 8, 0.99
    raw_rewards = (rng.normal(size=(T, B)) * 1.5).astype(np.float32)
    values = rng.normal(size=(T, B)).astype(np.float32) * 3
    dones = (rng.rand(T, B) < 0.15).astype(np.float32)
    boot = rng.normal(size=(B,)).astype(np.float32)

    class _L(object):
        rescale_reward = ActorLearner.rescale_reward
    rewards = np.zeros((T, B))
    for t in range(T):
        for b in range(B):
            rewards[t, b] = _L().rescale_reward(raw_rewards[t, b])
    # The n-step return loops themselves (paac.py:159-172, 351-365) are NOT transcribed here: they are pinned by
    # tests/golden/paac_loop.npz, captured from the unmodified train() loops (gen_golden_learner.py).
    rew_u = raw_rewards.astype(np.float64)
    # GAE as a3c/worker.py:284-300 for one column, lambda=0.96
    lam = 0.96
    col_r, col_v = rew_u[:, 0], np.concatenate([values[:, 0].astype(np.float64), [float(boot[0])]])
    deltas = np.array([col_r[t] + gamma * col_v[t + 1] - col_v[t] for t in range(T)])
    adv_gae = a3c_worker.GaussianWorker.gae_discount(deltas, gamma * lam)
    targets = adv_gae + col_v[:-1]
    disc_in = rng.normal(size=(17,))
    disc_out = a3c_worker.GaussianWorker.gae_discount(disc_in, 0.9504)

    class _LR(object):
        get_lr = ActorLearner.get_lr
    lr_steps = np.array([0, 1, 1000, 40000000, 80000000, 80000001, 10 ** 9])
    lrs = []
    for s in lr_steps:
        o = _LR(); o.global_step = int(s); o.lr_annealing_steps = 80000000; o.initial_lr = 1e-4
        lrs.append(o.get_lr())
    sx = np.concatenate([np.linspace(-50, 50, 101), [-745.0, 745.0, 0.0, -0.0]])
    save("returns", gamma=np.array(gamma), lam=np.array(lam), raw_rewards=raw_rewards, clipped_rewards=rewards,
         values=values, dones=dones, boot=boot, gae_deltas=deltas, gae_adv=adv_gae, gae_targets=targets,
         disc_in=disc_in, disc_out=disc_out, lr_steps=lr_steps, lrs=np.array(lrs),
         sigmoid_in=sx, sigmoid_out=a3c_worker.sigmoid(sx),
         sigmoid_scalar=np.array([a3c_worker.sigmoid(float(v)) for v in (-3.0, 0.0, 2.5)]))


if __name__ == "__main__":
    swarm_reset_fixture()
    swarm_step_fixture()
    swarm_traj_fixture()
    swarm_obs_fixture()
    swarm_action_fixture()
    solow_fixture()
    solow_runner_fixture()
    trade_fixture()
    returns_fixture()
