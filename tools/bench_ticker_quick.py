"""Ticker env alone: T-step rollout of scripted random actions, step + auto-reset + process_state
on the device, rewards captured and GAE(0.96) returns computed (a3c/worker.py:232-294).  Prints env-steps/s and the HBM
roofline of the step kernel.  Synthetic random-walk price table (no market data ships with the repo)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi  # noqa: E402
from goldsrl.envs.data.sampler import build_matrix  # noqa: E402

# algorithmic HBM bytes per env-step: actions 16; read+write cash 16, equity 16, positions 32, idx 8, elapsed 8, nhist 8;
# read start 4; write reward f32+f64 12, done 1, obs_raw 28, obs 28 (the 32-byte table rows are shared and cache-resident)
BYTES_PER_ENV_STEP = 177
HBM_PEAK_GBS = 8000.0

rng = np.random.RandomState(0)
days = 4096
opens = 100 * np.exp(np.cumsum(rng.normal(0, 0.006, days)))
closes = opens * np.exp(rng.normal(0, 0.006, days))
matrix = build_matrix(opens, closes, rng.uniform(1e5, 5e5, days).round())
T = 20
out = []
for E in (65536, 1048576):
    eng = _ffi.Engine(_ffi.ENV_TICKER, E, seed=1692)
    eng.ticker_set_table(matrix)
    eng.reset()
    acts = np.concatenate([rng.randint(0, 3, size=(T, E, 2)).astype(np.float32), rng.uniform(0, 1, size=(T, E, 2)).astype(np.float32)], axis=2)
    d_act = eng.dev_alloc(acts.nbytes)
    eng.dev_upload(d_act, acts)
    rew, val, boot, y, adv = (eng.dev_alloc(T * E * 4) for _ in range(5))
    rptr = eng.out_ptrs().reward

    def rollout():
        for t in range(T):
            eng.step_device(d_act + t * E * 16)
            eng.dev_copy(rew + t * E * 4, rptr, E * 4)
        eng.returns_device(rew, val, None, boot, T, E, 0.99, 0.96, 1.0, 0.0, 0.0, y, adv)

    rollout(); eng.wait()
    K = 10
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(K):
        rollout()
    eng.wait()
    dt = (time.perf_counter() - t0) / K
    launches, ms = eng.profile_read()
    eng.profile_enable(False)
    k_us = ms / max(launches, 1) * 1e3
    gbs = E * BYTES_PER_ENV_STEP / (k_us * 1e-6) / 1e9
    rec = {"workload": "TickerEnv, %d envs, T=%d, scripted actions, GAE returns" % (E, T), "env_steps_per_s": E * T / dt,
           "ms_per_rollout": dt * 1e3, "roofline": {"bound": "hbm", "kernel": "ticker_step_kernel", "avg_kernel_us": k_us, "launches": launches,
                                                     "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}}
    out.append(rec)
    print(json.dumps(rec), flush=True)
    eng.close()
