"""Which backward GEMM loses precision?  Reads the data gradients of the dense stack (one chunk) and compares them with the
float64 oracle's, fp16x3 form vs fp32 form (round 4 diagnosis; GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "golds-rl-gym_amd"), os.path.join(ROOT, "tests")]
from oracle import nets as NN
import test_gpu_net_tiles as T
from goldsrl import _ffi, _ffi_net

E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=21)
eng.reset()
rng = np.random.RandomState(100 + E)
lb, ab, pos = T._observations(eng, E, "interior", rng)
flat, p = T._biased_params(7)
states = T._states(lb, ab, pos)
n = E * 10
act = (rng.normal(size=(n, 2)) * 0.7).astype(np.float32)
adv = (rng.normal(size=n) * 0.02).astype(np.float32)
y = (-rng.rand(n) * 400).astype(np.float32)
mu, sigma, vs, c = NN.conv_forward(p, states, 1000.0, keep=True)
loss, pl, cl, dmu, dsigma, dvs = NN.gaussian_loss_terms(mu, sigma, act.astype(np.float64), adv.astype(np.float64), y.astype(np.float64), vs, 0.02, 1000.0)
dzmu = dmu * (1 - mu ** 2); dzsig = dsigma * sigma * (1 - sigma)
dp1 = (dzmu @ p["mu_w"].T + dzsig @ p["sigma_w"].T) * (c["p1"] > 0)
dzv = (dvs * (-1000.0) * NN._sigmoid(c["zv"]))[:, None]
dv2 = (dzv @ p["v3_w"].T) * (c["v2"] > 0)
dv1 = (dv2 @ p["v2_w"].T) * (c["v1"] > 0)
dd2 = (dp1 @ p["pol1_w"].T + dv1 @ p["v1_w"].T) * (c["d2"] > 0)
dd1 = (dd2 @ p["dense2_w"].T) * (c["d1"] > 0)
ref = {"gp1": dp1, "gv2": dv2, "gv1": dv1, "gd2": dd2, "gd1": dd1}
print("magnitudes (max |.|):", {k: "%.2e" % np.abs(v).max() for k, v in ref.items()})
print("policy part of dd2 vs value part: %.2e vs %.2e" % (np.abs(dp1 @ p["pol1_w"].T).max(), np.abs(dv1 @ p["v1_w"].T).max()))
for form in ("fp16x3", "f32"):
    if form == "f32":
        os.environ["GRL_NET_GEMM"] = "f32"
    net = _ffi_net.ConvNet(eng, max_chunk_samples=20000)
    net.set_params(flat)
    net.train_obs(lb, ab, pos, act, adv, y, lr=0.0, apply_update=False)
    for k, r in ref.items():
        got = net.read_activation(k, r.shape).astype(np.float64)
        m = np.abs(r) > 0.1 * np.abs(r).max()
        S = np.median(got[m] / r[m])
        S = 2.0 ** np.round(np.log2(S))
        d = np.abs(got / S - r)
        i = np.unravel_index(np.argmax(d), d.shape)
        mism = int(((got != 0) != (r != 0)).sum())
        print("%-7s %-4s loss scale 2^%-3d max err / max %.2e  rms err / rms %.2e   worst at %s (ref %.3e got %.3e)  mask mismatches %d" %
              (form, k, int(np.log2(S)), d.max() / np.abs(r).max(), np.sqrt((d ** 2).mean()) / np.sqrt((r ** 2).mean()), i, r[i], got[i] / S, mism))
    net.close()
