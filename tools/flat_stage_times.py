"""Stage clock of the persistent flat rollout: where one workgroup's time goes (workgroup 0, 100 MHz constant clock)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))
from goldsrl import _ffi  # noqa: E402
from goldsrl import rollout as R  # noqa: E402

kind, E = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("solow", 4096)
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
bwd = len(sys.argv) > 3 and sys.argv[3] == "bwd"
if graph:
    os.environ["GRL_FLAT_ROLLOUT"] = "graph"
T = 20
if kind == "solow":
    eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=1692)
else:
    eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=1692, n_assets=16, rnn_length=20)
eng.reset()
roll = R.FlatPolicyRollout(eng, T, train=False)
if bwd:      # stage clock of the gradient step's backward kernel (workgroup 0, its first group)
    roll.run(); eng.wait()
    roll.net.rollout_stage_times()
    roll.net.train_rollout_grads()
    ts = roll.net.rollout_stage_times().astype(np.float64) * 0.01
    rnn = roll.net.cfg.rnn_length
    nf = 2 * rnn + 7                 # the training forward stamps first
    d = np.diff(ts)
    print("stamps", len(ts))
    nb = 8 + 3 * rnn + 1             # per group: entry + St0..St5 boundaries + 3 per GRU step + end
    k = None
    # the forward kernel's workgroup 0 wrote nf stamps; then the backward's groups follow
    seg = d[nf:nf + nb]
    print("backward, first group, stages (us):", np.round(seg, 2).tolist(), "sum %.1f" % float(np.sum(seg)))
    sys.exit(0)
roll.net.rollout_stage_times()      # attach before the first rollout (the graph path captures its kernel arguments then)
roll.run(); eng.wait()
if not graph:
    roll.run(); eng.wait()
ts = roll.net.rollout_stage_times().astype(np.float64) * 0.01      # us
d = np.diff(ts)
rnn = roll.net.cfg.rnn_length
per_fwd = 2 * rnn + 6          # stamps inside one forward (entry + P0 + 2 per GRU step + DT + H1 + H2 + H3)
print("stamps", len(ts), "total us", ts[-1] - ts[0])
if graph:      # the forward kernel alone: only the LAST launch's stamps survive a rollout? no: the counter runs on; show one forward
    per = 2 * rnn + 7
    k = per * 3
    print("standalone forward stages (us):", np.round(d[k:k + per - 1], 2).tolist(), "sum %.2f" % float(np.sum(d[k:k + per - 1])))
    sys.exit(0)
# layout: [start], then per step: [before fwd], fwd stamps..., [after sample], [after env], [after prices/tape]
i = 1
for t in range(2):
    seg = d[i:i + per_fwd + 3 + 1]
    print("step", t, "record->fwd0 %.2f" % seg[0], "fwd stages:", np.round(seg[1:per_fwd + 1], 2).tolist(),
          "sample %.2f env %.2f post %.2f" % tuple(seg[per_fwd + 1:per_fwd + 4]))
    i += per_fwd + 4
print("forward total per step (us):", float(np.sum(d[2:2 + per_fwd])))
