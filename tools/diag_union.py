"""Which gradient blocks differ between GRL_TRUNK_SKIP on / off?  (inputs of test_trunk_row_lists_change_nothing_but_the_work[strip])"""
import os, sys, numpy as np
sys.path.insert(0, 'golds-rl-gym_amd'); sys.path.insert(0, '.')
from goldsrl import _ffi, _ffi_net
from oracle import nets as NN
E = 90
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=3); eng.reset()
rng = np.random.RandomState(15)
lb = rng.randint(0, 84, size=(E, 80, 2)).astype(np.uint8)
pos = rng.randint(0, 84, size=(E, 10, 2)).astype(np.uint8)
for e in range(E):
    if e % 3 == 0: lb[e] = rng.randint(0, 12, size=(80, 2))
    if e % 7 == 0: lb[e] = 255; pos[e, :2] = 255
    if e % 5 == 0: pos[e, 2:6, 0] = rng.choice([0, 1, 82, 83], size=4)
lb[lb[:, :, 0] != 255, 1] %= 22
pos[pos[:, :, 0] != 255, 1] %= 22
ab = pos.copy()
r2 = np.random.RandomState(16); n = E * 10
act = r2.normal(size=(n, 2)).astype(np.float32) * 0.7; adv = (r2.normal(size=n) * 0.02).astype(np.float32); y = (-r2.rand(n) * 400).astype(np.float32)
flat = _ffi_net.glorot_uniform_flat(seed=17).astype(np.float64)
p = NN.unflatten_params(flat)
for k in p:
    if k.endswith("_b"): p[k] = rng.normal(size=p[k].shape) * 0.05
flat = NN.flatten_params(p).astype(np.float32)
res = {}
for mode in ("on", "off"):
    os.environ["GRL_TRUNK_SKIP"] = mode
    net = _ffi_net.ConvNet(eng, max_chunk_samples=int(os.environ.get("CHUNK", "370")))
    net.set_params(flat)
    net.train_obs(lb, ab, pos, act, adv, y, lr=0.0, apply_update=False)
    res[mode] = NN.unflatten_params(net.get_grads().astype(np.float64), NN.CONV_PARAM_SHAPES)
    net.close()
for name, _ in NN.CONV_PARAM_SHAPES:
    a, b = res["on"][name], res["off"][name]
    d = np.abs(a - b); m = np.abs(b).max()
    print("%-10s max rel %.3g  elements > 1e-5 of max: %d of %d" % (name, d.max() / m, int((d > 1e-5 * m).sum()), d.size))
    if name == "dense1_w" and d.max() > 1e-5 * m:
        rows = np.nonzero((d > 1e-5 * m).any(axis=1))[0]
        print("   rows (pixel = row // 64):", sorted(set((rows // 64).tolist())))
