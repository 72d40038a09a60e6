import os, sys
import numpy as np
sys.path.insert(0, 'golds-rl-gym_amd'); sys.path.insert(0, '.')
from goldsrl import _ffi, _ffi_net
from oracle import oracle as O
E = 512
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=21); eng.reset()
rng = np.random.RandomState(0)
for _ in range(2):
    eng.step(O.swarm_transform_actions(rng.normal(size=(E, 10, 2)).astype(np.float32)))
flat = _ffi_net.glorot_uniform_flat(seed=3)
flat = flat + (rng.normal(size=flat.size) * 0.01).astype(np.float32)
outs = {}
for mode in ("lds", "gemm"):
    os.environ["GRL_NET_EXPAND2"] = mode
    net = _ffi_net.ConvNet(eng, max_chunk_samples=E * 10, reserved=4)
    net.set_params(flat)
    p = net.predict()
    a2 = net.read_activation("a2", (E * 10, 9, 9, 64))
    outs[mode] = (p, a2)
    net.close()
a, b = outs["lds"][1], outs["gemm"][1]
d = np.abs(a - b)
print("a2 max abs diff", d.max(), "rel to max", d.max() / np.abs(a).max(), "mismatching sign", ((a > 0) != (b > 0)).sum(), "of", a.size)
idx = np.argwhere(d > 1e-4)
print("elements > 1e-4:", len(idx), idx[:5])
for k in ("mu", "sigma", "vs"):
    print(k, np.abs(outs["lds"][0][k] - outs["gemm"][0][k]).max())
