#!/usr/bin/env python3
"""Golden vectors for the env constructor / step kwargs the facades accept since round 4, captured from the UNMODIFIED reference
(/root/reference, read-only) under the gym/tensorflow stand-ins of _ref_stubs.py:

  * SwarmEnv._step(v_action, add_wind=False)           (envs/multiagent.py:30-44), float64 and float32 action rows
  * TradeAR1Env(starting_balance, n_assets, std_p)      (envs/fed_env.py:269-330): reset observation, scripted steps with the
    price draws captured, and a depletion below MIN_CASH so that the worker-style auto-reset shows the starting balance again

Run in the build container only:   python tests/golden/gen_golden_kwargs.py   -> env_kwargs.npz
Fixtures are data (inputs + expected outputs); no reference source text is stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()

import fed_gym  # noqa: E402,F401
from fed_gym.envs import fed_env, multiagent  # noqa: E402


def swarm_nowind():
    rng = np.random.RandomState(4242)
    out = {k: [] for k in ("x", "xa", "action", "agent_noise", "particle_noise", "x_out", "xa_out", "reward", "f32")}
    for seed in (192, 11):
        env = multiagent.SwarmEnv(seed=seed)
        env.reset()
        for step in range(24):
            f32 = step % 2 == 1
            action = rng.normal(size=(10, 2)) * (0.0 if step % 7 == 3 else 1.0)
            if f32:
                action = action.astype(np.float32)      # what the worker reads from the learner's shared c_float array
            x_in, xa_in = env.states[0].copy(), env.states[1].copy()
            a_row, p_row = env.agent_noise[env.t].copy(), env.particle_noise[env.t].copy()
            (x, xa), r, d, _ = env._step(action, add_wind=False)
            if step % 3 != 2:
                out["x"].append(x_in); out["xa"].append(xa_in); out["action"].append(action.astype(np.float64))
                out["agent_noise"].append(a_row); out["particle_noise"].append(p_row)
                out["x_out"].append(x.copy()); out["xa_out"].append(xa.copy()); out["reward"].append(r); out["f32"].append(f32)
    return {"nowind_" + k: np.array(v) for k, v in out.items()}


def trade_kwargs():
    out = {}
    n, sb, sp = 3, 25.0, 0.1
    env = fed_env.TradeAR1Env(starting_balance=sb, n_assets=n, std_p=sp)
    obs0 = env.reset()
    steps = 32
    np.random.seed(99)
    normals = np.random.normal(size=(steps, n))      # what _price_transition will draw
    np.random.seed(99)
    acts = np.tanh(np.random.RandomState(5).normal(size=(steps, n)))
    acts[4] = 0.0; acts[5] = -1.0; acts[6] = 1.0
    obs, rew, done = [], [], []
    for t in range(steps):
        o, r, d, _ = env.step(acts[t].copy())
        obs.append(o); rew.append(r); done.append(d)
    out.update(tk_n=np.array(n), tk_starting_balance=np.array(sb), tk_std_p=np.array(sp), tk_std_e=np.array(env.std_e), tk_obs0=obs0,
               tk_normals=normals, tk_actions=acts, tk_obs=np.array(obs), tk_reward=np.array(rew), tk_done=np.array(done))
    # depletion from a small starting balance: buy everything, prices forced down until assets < MIN_CASH
    env = fed_env.TradeAR1Env(starting_balance=3.0, n_assets=2)
    obs0 = env.reset()
    np.random.seed(6)
    normals = -np.abs(np.random.normal(size=(200, 2))) * 40.0
    it = iter(normals)
    orig = np.random.normal
    np.random.normal = lambda size=None: next(it)
    obs, rew, done, acts = [], [], [], []
    try:
        for t in range(200):
            a = np.array([1.0, 1.0]) if t < 3 else np.array([0.0, 0.0])
            o, r, d, _ = env.step(a.copy())
            obs.append(o); rew.append(r); done.append(d); acts.append(a)
            if d:
                break
    finally:
        np.random.normal = orig
    assert done[-1]
    out.update(td_obs0=obs0, td_reset_obs=env.reset(), td_normals=normals[:len(obs)], td_actions=np.array(acts), td_obs=np.array(obs),
               td_reward=np.array(rew), td_done=np.array(done))
    return out


if __name__ == "__main__":
    arrays = {}
    arrays.update(swarm_nowind())
    arrays.update(trade_kwargs())
    path = os.path.join(HERE, "env_kwargs.npz")
    np.savez_compressed(path, **arrays)
    print("wrote env_kwargs.npz %d bytes keys=%s" % (os.path.getsize(path), sorted(arrays)))
