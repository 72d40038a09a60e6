#!/usr/bin/env python3
"""Golden fixture for SolowSSEnv / `SolowSS-v0` (reference fed_gym/envs/fed_env.py:253-265, fed_gym/__init__.py:15-19), made by
running the UNMODIFIED reference under the stand-ins of _ref_stubs.py.  Build container only:  python tests/golden/gen_golden_solowss.py

SolowSSEnv._reset ends with np.array([self.k, self.z]).flatten() where z is a (1,) array: numpy 1.13 (the reference's pin)
flattens that to [k, 0.], numpy >= 1.24 (this image) raises ValueError AFTER every field has been assigned (fed_env.py:258-263).
The generator therefore takes the reset observation from the fields it just checked ([k, z[0]]) and carries on stepping."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()
import gym  # noqa: E402
import fed_gym  # noqa: E402,F401
from fed_gym.envs import fed_env  # noqa: E402


def main():
    out = {}
    entry, cap, kwargs = gym.envs.registry["SolowSS-v0"]
    assert entry == "fed_gym.envs:SolowSSEnv" and cap == 1024 and kwargs == {}
    env = fed_env.SolowSSEnv()
    assert (env.sigma, env.p, env.q, env.delta, env.T) == (0.02, 1, 0, 0.02, 2048)
    np.random.seed(1692)
    try:
        obs0 = env.reset()
    except ValueError:
        obs0 = np.array([env.k, env.z[0]])
    assert env.k == env._k_ss(env.alpha) and np.array_equal(env.z, [0.]) and env.e == 0.
    tape = np.array(env.es)
    rng = np.random.RandomState(9)
    s_seq = np.concatenate([np.full(6, 0.33), rng.rand(24), [0.0, 1.0]])
    obs, rew = [], []
    for s in s_seq:
        o, r, d, _ = env.step(float(s))
        assert d is False
        obs.append(o); rew.append(r)
    out.update(obs0=obs0, tape_tail=tape[-64:], s=s_seq, obs=np.array(obs), reward=np.array(rew), rho_z=np.asarray(env.rho_z),
               rho_e=np.asarray(env.rho_e), sigma=np.array(env.sigma), max_episode_steps=np.array(cap))
    path = os.path.join(HERE, "solow_ss.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
