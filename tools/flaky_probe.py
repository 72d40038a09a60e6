"""Run the resident-vs-recompute comparison several times in one process and print where results differ."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "golds-rl-gym_amd"))
import test_gpu_net as T
from goldsrl import _ffi_net

names = []
for name, shape in _ffi_net.CONV_PARAM_SHAPES:
    names += [name] * int(np.prod(shape))
names = np.array(names)

for rep in range(6):
    out = []
    for flags in (0, 2, 0):
        eng, net, p, states, obs = T._setup(12, flags=flags)
        net.rollout(3, 0)
        eng.wait()
        st = net.train_rollout(1e-3)
        out.append((net.get_grads().copy(), net.read_rollout("values", (3, 120)).copy(), st))
    for a, b, tag in ((0, 1, "keep-vs-recompute"), (0, 2, "keep-vs-keep")):
        g0, g1 = out[a][0], out[b][0]
        bad = np.nonzero(g0 != g1)[0]
        vals_equal = np.array_equal(out[a][1], out[b][1])
        msg = "rep %d %s: values equal %s, grads differing %d" % (rep, tag, vals_equal, bad.size)
        if bad.size:
            u, c = np.unique(names[bad], return_counts=True)
            rel = np.abs(g0[bad] - g1[bad]) / np.maximum(np.abs(g0[bad]), 1e-30)
            msg += " in %s, max rel %.3g, stats %s vs %s" % (dict(zip(u, c)), rel.max(), out[a][2], out[b][2])
        print(msg, flush=True)
