"""Diagnostic for tests/test_gpu_net.py::test_gradient_paths_agree_at_tile_multiple_sizes: per-layer max difference (relative to the
layer's largest gradient) between the shared-trunk evaluation and the per-agent evaluation, for both conv2-correction kernels."""
import os, sys
import numpy as np
sys.path.insert(0, 'golds-rl-gym_amd'); sys.path.insert(0, '.')
from goldsrl import _ffi, _ffi_net
from oracle import nets as NN, oracle as O
E = 256
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=21); eng.reset()
rng = np.random.RandomState(0)
for _ in range(2):
    eng.step(O.swarm_transform_actions(rng.normal(size=(E, 10, 2)).astype(np.float32)))
obs = (eng.read("locust_bins"), eng.read("agent_bins"), eng.read("positions"))
flat = _ffi_net.glorot_uniform_flat(seed=3)
flat = flat + (rng.normal(size=flat.size) * 0.01).astype(np.float32)
r2 = np.random.RandomState(5); n = E * 10
act, adv, y = r2.normal(size=(n, 2)).astype(np.float32) * 0.7, (r2.normal(size=n) * 0.02).astype(np.float32), (-r2.rand(n) * 400).astype(np.float32)
def grads(chunk, flags, mode):
    os.environ["GRL_NET_EXPAND2"] = mode
    net = _ffi_net.ConvNet(eng, max_chunk_samples=chunk, reserved=flags)
    net.set_params(flat); net.train_obs(*obs, act, adv, y, lr=0.0, apply_update=False)
    g = NN.unflatten_params(net.get_grads().astype(np.float64)); net.close(); return g
ref = grads(640, 1, "lds")
for mode in ("lds", "gemm"):
    for chunk in (2560, 1280):
        g = grads(chunk, 0, mode)
        out = []
        for name in ("conv1_w", "conv2_w", "conv3_w", "dense1_w", "dense2_w"):
            d = np.abs(g[name] - ref[name]); s = np.abs(ref[name]).max()
            out.append("%s max %.2e n>1e-4: %d" % (name, d.max() / s, (d / s > 1e-4).sum()))
        print(mode, chunk, " | ".join(out))
