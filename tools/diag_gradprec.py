"""Per-block error of the gradient against the float64 oracle: fp16x3 form vs fp32 form, with / without the loss scale, and with
value-head gradients of different size relative to the policy head's (round 4 diagnosis; GPU box).
usage: python tools/diag_gradprec.py [E] [regime]"""
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
adv0 = (rng.normal(size=n) * 0.02).astype(np.float32)
y0 = (-rng.rand(n) * 400).astype(np.float32)
mu, sigma, vs = NN.conv_forward(p, states, 1000.0)
names = [k for k, _ in NN.CONV_PARAM_SHAPES]


def run(tag, adv, y, env=None, **kw):
    _, _, _, g, _ = NN.conv_loss_and_grads(p, states, act.astype(np.float64), adv.astype(np.float64), y.astype(np.float64), 0.02, 1000.0)
    for k, v in (env or {}).items():
        os.environ[k] = v
    net = _ffi_net.ConvNet(eng, max_chunk_samples=500, **kw)
    net.set_params(flat)
    net.train_obs(lb, ab, pos, act, adv, y, lr=0.0, apply_update=False)
    got = NN.unflatten_params(net.get_grads().astype(np.float64))
    net.close()
    for k in (env or {}):
        del os.environ[k]
    print("%-44s" % tag, " ".join("%s=%.0e" % (k.replace("_w", "W").replace("_b", "B").replace("conv", "c").replace("dense", "d"),
                                                 np.abs(got[k] - g[k]).max() / np.abs(g[k]).max()) for k in names))


print("value-head gradient ~ (vs - y): vs in [%.0f, %.0f], y in [%.0f, %.0f]" % (vs.min(), vs.max(), y0.min(), y0.max()))
run("fp16x3", adv0, y0)
run("f32", adv0, y0, {"GRL_NET_GEMM": "f32"})
run("fp16x3, loss scale off", adv0, y0, {"GRL_NET_LOSS_SCALE": "off"})
ynear = (vs + rng.normal(size=n) * 1.0).astype(np.float32)      # targets near the values: critic gradient ~ the policy's
run("fp16x3, y = vs + N(0,1)", adv0, ynear)
run("f32,    y = vs + N(0,1)", adv0, ynear, {"GRL_NET_GEMM": "f32"})
run("fp16x3, adv x 1000", (adv0 * 1000).astype(np.float32), y0)
run("f32,    adv x 1000", (adv0 * 1000).astype(np.float32), y0, {"GRL_NET_GEMM": "f32"})
yeq = vs.astype(np.float32)
run("fp16x3, y = vs (policy only)", adv0, yeq)
run("f32,    y = vs (policy only)", adv0, yeq, {"GRL_NET_GEMM": "f32"})
