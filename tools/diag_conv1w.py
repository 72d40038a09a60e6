"""Where does conv1_w's gradient differ from the float64 oracle at 1 280 samples?  (round 4 diagnosis; GPU box)
usage: python tools/diag_conv1w.py [E] [regime]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "golds-rl-gym_amd"), os.path.join(ROOT, "tests")]
from oracle import nets as NN
import test_gpu_net_tiles as T
from goldsrl import _ffi, _ffi_net

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
regime = sys.argv[2] if len(sys.argv) > 2 else "interior"
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=21)
eng.reset()
rng = np.random.RandomState(100 + E)
lb, ab, pos = T._observations(eng, E, regime, rng)
flat, p = T._biased_params(7)
states = T._states(lb, ab, pos)
n = E * 10
act = (rng.normal(size=(n, 2)) * 0.7).astype(np.float32)
adv = (rng.normal(size=n) * 0.02).astype(np.float32)
y = (-rng.rand(n) * 400).astype(np.float32)
loss, pl, cl, g, cache = NN.conv_loss_and_grads(p, states, act.astype(np.float64), adv.astype(np.float64), y.astype(np.float64), 0.02, 1000.0)
ref = g["conv1_w"]
print("ref conv1_w max per input channel:", [float(np.abs(ref[:, :, c]).max()) for c in range(3)])


def run(tag, env=None, **kw):
    for k, v in (env or {}).items():
        os.environ[k] = v
    net = _ffi_net.ConvNet(eng, **kw)
    net.set_params(flat)
    net.train_obs(lb, ab, pos, act, adv, y, lr=0.0, apply_update=False)
    got = NN.unflatten_params(net.get_grads().astype(np.float64))
    net.close()
    for k in (env or {}):
        del os.environ[k]
    d = np.abs(got["conv1_w"] - ref)
    scale = np.abs(ref).max()
    idx = np.unravel_index(np.argsort(d.reshape(-1))[::-1][:5], d.shape)
    worst = {k: float(np.abs(got[k] - g[k]).max() / np.abs(g[k]).max()) for k in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "conv3_w", "dense1_w")}
    print("%-34s conv1_w err %.2e  per channel %s" % (tag, d.max() / scale, ["%.1e" % (d[:, :, c].max() / scale) for c in range(3)]))
    print("      top entries (ky,kx,c,co):", [tuple(int(i[j]) for i in idx) + ("%.2e" % d[tuple(i[j] for i in idx)],) for j in range(3)])
    print("      other blocks:", {k: "%.1e" % v for k, v in worst.items()})
    return got


a = run("default chunk 500", max_chunk_samples=500)
run("chunk 640", max_chunk_samples=640)
run("one chunk", max_chunk_samples=20000)
run("chunk 250", max_chunk_samples=250)
run("TRUNK_SKIP off", {"GRL_TRUNK_SKIP": "off"}, max_chunk_samples=500)
run("PATCH_SKIP off", {"GRL_PATCH_SKIP": "off"}, max_chunk_samples=500)
run("both off", {"GRL_TRUNK_SKIP": "off", "GRL_PATCH_SKIP": "off"}, max_chunk_samples=500)
run("per-agent trunk", max_chunk_samples=500, reserved=1)
run("f32 gemms", {"GRL_NET_GEMM": "f32"}, max_chunk_samples=500)
run("one lane", {"GRL_NET_LANES": "1"}, max_chunk_samples=500)
# how close to zero do conv1 pre-activations come?  (a ReLU mask that flips between float32 and float64 moves whole gradient entries)
z1, _ = NN._conv(states, p["conv1_w"], p["conv1_b"], 4)
for eps in (1e-5, 1e-6, 1e-7, 1e-8):
    print("conv1 pre-activations within %.0e of zero: %d of %d" % (eps, int((np.abs(z1) < eps).sum()), z1.size))
