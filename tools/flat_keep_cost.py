"""What keeping the activations costs the rollout and saves the gradient step: ms per rollout (keep off / on), ms per gradient step (forward + backward / backward only)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))
from goldsrl import _ffi  # noqa: E402
from goldsrl import rollout as R  # noqa: E402

T = 20
for kind, E in (("solow", 4096), ("trade", 8192)):
    if kind == "solow":
        eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=1692)
    else:
        eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=1692, n_assets=16, rnn_length=20)
    eng.reset()
    roll = R.FlatPolicyRollout(eng, T, train=False)
    row = []
    for keep in (0, 1):
        roll.net.set_keep_activations(keep)
        for _ in range(3):
            roll.net.rollout(T); eng.wait()
        best_r, best_g = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                roll.net.rollout(T); eng.wait()
            best_r.append((time.perf_counter() - t0) / 10 * 1e3)
            roll.net.rollout(T); eng.wait()
            t0 = time.perf_counter()
            for _ in range(10):
                roll.net.train_rollout_grads()      # parameters do not move: a kept workspace stays valid
            best_g.append((time.perf_counter() - t0) / 10 * 1e3)
        row.append("keep=%d rollout %.3f gradient step %.3f" % (keep, min(best_r), min(best_g)))
    print(kind, E, "  ".join(row), flush=True)
    roll.net.close(); eng.close()
