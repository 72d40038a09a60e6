import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))
from goldsrl import _ffi, _ffi_net
E, T = 32768, 2
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=78)
eng.reset()
net = _ffi_net.ConvNet(eng)
net.set_params(_ffi_net.glorot_uniform_flat(seed=9))
net.rollout(T, 0)
gs = []
for i in range(4):
    net.train_rollout(0.0)
    gs.append(net.get_grads().copy())
for i in range(1, 4):
    d = np.abs(gs[i] - gs[0])
    nz = np.nonzero(d)[0]
    print(i, "differs at", len(nz), "of", len(d), "max abs", d.max(), "first", nz[:5], "last", nz[-5:] if len(nz) else None,
          "max rel", (d[nz] / np.maximum(np.abs(gs[0][nz]), 1e-30)).max() if len(nz) else 0)


blocks = [("c1w", 6144, 32), ("c1b", 32, 32), ("c2w", 32768, 64), ("c2b", 64, 64), ("c3w", 36864, 64), ("c3b", 64, 64), ("d1w", 1605632, 512), ("d1b", 512, 512),
          ("d2w", 131072, 256), ("d2b", 256, 256), ("p1w", 131072, 512), ("p1b", 512, 512)]
off = 0
d = gs[1] != gs[0]
for name, cnt, cols in blocks:
    dd = d[off:off + cnt].reshape(-1, cols)
    rows, cs = np.nonzero(dd)
    print(name, "differs", dd.sum(), "of", cnt, "cols%4 histogram", np.bincount(cs % 4, minlength=4).tolist(), "cols<256:", int((cs < 256).sum()), "rows", np.unique(rows)[:6])
    off += cnt
print("tail (heads etc.) differs", d[off:].sum(), "of", len(d) - off)
