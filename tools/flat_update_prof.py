"""flat PAAC updates (rollout + gradient step) for a rocprofv3 --kernel-trace --stats pass"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))
from goldsrl import _ffi  # noqa: E402
from goldsrl import rollout as R  # noqa: E402
kind, E = sys.argv[1], int(sys.argv[2])
eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=1692) if kind == "solow" else _ffi.Engine(_ffi.ENV_TRADE, E, seed=1692, n_assets=16, rnn_length=20)
eng.reset()
roll = R.FlatPolicyRollout(eng, 20, train=True)
for _ in range(10):
    roll.run()
eng.wait()
