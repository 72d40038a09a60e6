"""Persistent flat rollout: ms per rollout by envs per workgroup (GRL_FLAT_GROUP = 64 / 32 / 16; 0 = the product's choice)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))
from goldsrl import _ffi  # noqa: E402
from goldsrl import rollout as R  # noqa: E402

T = 20
cases = [("solow", 4096), ("solow", 8192), ("trade", 4096), ("trade", 8192), ("trade", 16384)]
if len(sys.argv) > 2:
    cases = [(sys.argv[1], int(sys.argv[2]))]
for kind, E in cases:
    row = []
    for g in (64, 32, 16, 0):
        if g:
            os.environ["GRL_FLAT_GROUP"] = str(g)
        else:
            os.environ.pop("GRL_FLAT_GROUP", None)
        if kind == "solow":
            eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=1692)
        else:
            eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=1692, n_assets=16, rnn_length=20)
        eng.reset()
        roll = R.FlatPolicyRollout(eng, T, train=False)
        for _ in range(3):
            roll.run(); eng.wait()
        best = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                roll.run(); eng.wait()
            best.append((time.perf_counter() - t0) / 10 * 1e3)
        row.append("G=%s %.3f" % (g or "auto", min(best)))
        roll.net.close(); eng.close()
    print("%s E=%d  ms per rollout (T=%d, best of 3x10):  " % (kind, E, T) + "   ".join(row), flush=True)
