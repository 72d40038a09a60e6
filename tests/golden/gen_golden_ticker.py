#!/usr/bin/env python3
"""Golden vectors for the Ticker path, captured from the UNMODIFIED reference
(/root/reference, read-only) under the gym/tensorflow stand-ins of _ref_stubs.py:

  * OpenCloseSampler.open_close_to_sequence / _get_inverse   (fed_gym/envs/data/sampler.py:15-41)
  * TickerEnv._reset / _step                                 (fed_gym/envs/fed_env.py:110-158)
  * TickerTraderStateProcessor.process_state / process_temporal_states (fed_gym/agents/state_processors.py:45-66)
  * TickerGatedTraderWorker.transform_raw_action             (fed_gym/agents/a3c/worker.py:491-494)

Run in the build container only:   python tests/golden/gen_golden_ticker.py
Two things cannot run as shipped and are fed through stand-ins that carry no arithmetic:
  - sampler.py:18 indexes a pandas Series with [:, None], which pandas >= 2 refuses: the price table is handed to the
    unmodified method as a dict of ndarray columns that answer `.values` / `.iloc` (numpy semantics, what pandas 0.x did);
  - TickerEnv.__init__ reads IEF.csv through that sampler: the env object is created without __init__, given the
    attributes __init__ sets (fed_env.py:93-108) and a sampler stand-in whose sample(n) returns scripted windows of a
    synthetic matrix (random.randint of sampler.py:38 is not reproducible on a device anyway: the window start is an input).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()

from fed_gym.envs import fed_env  # noqa: E402
from fed_gym.envs.data import sampler as ref_sampler  # noqa: E402
from fed_gym.agents.state_processors import TickerTraderStateProcessor  # noqa: E402
from fed_gym.agents.a3c import worker as a3c_worker  # noqa: E402


class _Col(np.ndarray):
    """ndarray column that answers the two pandas accessors the sampler uses."""
    values = property(lambda self: np.asarray(self))
    iloc = property(lambda self: np.asarray(self))


def _col(a):
    return np.asarray(a, dtype=np.float64).view(_Col)


class _Windows(object):
    """sampler stand-in: sample(n) returns matrix[start:start+n] for the scripted starts, in order."""

    def __init__(self, matrix, starts):
        self.matrix, self.starts, self.served = matrix, list(starts), []

    def sample(self, n):
        s = self.starts.pop(0)
        self.served.append(s)
        return self.matrix[s:s + n]


def synthetic_table(days, seed, crash_at=None):
    rng = np.random.RandomState(seed)
    opens, closes = [], []
    p = 100.0
    for d in range(days):
        o = p * np.exp(rng.normal(0, 0.004))
        c = o * np.exp(rng.normal(0, 0.006))
        if crash_at is not None and d >= crash_at:
            c = o * 0.55          # prices collapse: equity falls below MIN_CASH for a fully invested account
        opens.append(o); closes.append(c)
        p = c
    vol = rng.uniform(1e5, 5e5, size=days).round()
    return np.array(opens), np.array(closes), vol


def make_env(matrix, starts):
    env = object.__new__(fed_env.TickerEnv)          # fed_env.py:93-108 without the CSV read
    env.MIN_CASH = 1.
    env.starting_balance = 10.
    env.n_assets = 2
    env.cash_balance = env.prices = env.assets = env.quantities = None
    env.spread = 0.006
    env.data = _Windows(matrix, starts)
    env.data_idx = None
    return env


def run_episode(env, disc, cont):
    obs0 = env.reset()
    obs, rew, done, cash, qty, assets = [], [], [], [], [], []
    for t in range(len(disc)):
        o, r, d, _ = env.step([disc[t].copy(), cont[t].copy()])      # _step rescales the caller's buy fractions in place
        obs.append(o); rew.append(r); done.append(d)
        cash.append(env.cash_balance); qty.append(env.quantities.copy()); assets.append(env.assets)
        if d:
            break
    return obs0, np.array(obs), np.array(rew), np.array(done), np.array(cash), np.array(qty), np.array(assets)


def main():
    out = {}
    # ---- sampler
    s = object.__new__(ref_sampler.OpenCloseSampler)
    opens, closes, vol = synthetic_table(700, 11)
    frame = {"Open": _col(opens), "Close": _col(closes), "Volume": _col(vol)}
    matrix = s.open_close_to_sequence(frame, inverse_asset=True)
    out["tbl_open"], out["tbl_close"], out["tbl_volume"], out["matrix"] = opens, closes, vol, matrix
    # ---- env, scripted mixed actions on several windows
    rng = np.random.RandomState(3)
    starts = [0, 17, 123, 376]
    T = 48
    for e, st in enumerate(starts):
        env = make_env(matrix, [st])
        disc = rng.randint(0, 3, size=(T, 2))
        cont = 1.0 / (1.0 + np.exp(-rng.normal(size=(T, 2))))          # what transform_raw_action hands over: sigmoid
        if e == 0:
            disc[0] = [1, 1]; cont[0] = [0.9, 0.8]                       # buy fractions summing above 1: rescaled
            disc[1] = [2, 2]; cont[1] = [1.0, 1.0]                       # sell everything
            disc[2] = [0, 0]                                             # hold
            disc[3] = [2, 1]; cont[3] = [0.5, 0.3]                       # sell an empty position, buy the other
        o0, o, r, d, c, q, a = run_episode(env, disc, cont)
        k = "e%d_" % e
        out[k + "start"], out[k + "disc"], out[k + "cont"] = np.array(st), disc, cont
        out[k + "obs0"], out[k + "obs"], out[k + "reward"], out[k + "done"] = o0, o, r, d
        out[k + "cash"], out[k + "qty"], out[k + "assets"] = c, q, a
    # ---- env reaching done: fully invested into a collapsing price column (the matrix is written directly: the
    # sampler's own exact-zero self-check, sampler.py:36, does not survive such returns in floating point)
    n = 80
    px = 100.0 * np.concatenate([np.ones(6), 0.55 ** np.arange(1, n - 5)])
    crash = np.stack([px, 100.0 * 100.0 / px, np.zeros(n), np.zeros(n)], axis=1)
    out["crash_matrix"] = crash
    env = make_env(crash, [0])
    disc = np.zeros((60, 2), dtype=np.int64); cont = np.zeros((60, 2))
    disc[0] = [1, 0]; cont[0] = [1.0, 0.0]
    o0, o, r, d, c, q, a = run_episode(env, disc, cont)
    assert d[-1], "crash case must reach done"
    out["crash_disc"], out["crash_cont"] = disc[:len(o)], cont[:len(o)]
    out["crash_obs0"], out["crash_obs"], out["crash_reward"], out["crash_done"] = o0, o, r, d
    out["crash_cash"], out["crash_qty"], out["crash_assets"] = c, q, a
    # ---- observation / action transforms
    proc = TickerTraderStateProcessor(2)
    raw = out["e1_obs"][:10]
    out["proc_in"] = raw
    out["proc_out"] = np.array([proc.process_state(x) for x in raw])
    out["proc_temporal"] = proc.process_temporal_states([proc.process_state(x) for x in raw[:5]])
    x = np.linspace(-4, 4, 17)
    dsc, cnt = a3c_worker.TickerGatedTraderWorker.transform_raw_action(None, np.array([1, 2, 0]), x)
    out["tra_in"], out["tra_out"] = x, cnt
    path = os.path.join(HERE, "ticker.npz")
    np.savez_compressed(path, **out)
    print("wrote ticker.npz %d bytes, keys=%d" % (os.path.getsize(path), len(out)))


if __name__ == "__main__":
    main()
