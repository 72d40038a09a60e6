import os, sys, numpy as np
sys.path.insert(0, 'golds-rl-gym_amd'); sys.path.insert(0, '.')
from goldsrl import _ffi, _ffi_net
from oracle import nets as NN
from oracle import oracle as O
E = 256
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=21); eng.reset()
rng = np.random.RandomState(0)
for _ in range(2): eng.step(O.swarm_transform_actions(rng.normal(size=(E, 10, 2)).astype(np.float32)))
obs = (eng.read("locust_bins"), eng.read("agent_bins"), eng.read("positions"))
flat = _ffi_net.glorot_uniform_flat(seed=3); flat = flat + (rng.normal(size=flat.size) * 0.01).astype(np.float32)
rng2 = np.random.RandomState(5); n = E * 10
act = rng2.normal(size=(n, 2)).astype(np.float32) * 0.7; adv = (rng2.normal(size=n) * 0.02).astype(np.float32); y = (-rng2.rand(n) * 400).astype(np.float32)
res = {}
for mode in ("on", "off"):
    os.environ["GRL_TRUNK_SKIP"] = mode
    net = _ffi_net.ConvNet(eng, max_chunk_samples=2560)
    net.set_params(flat)
    net.train_obs(*obs, act, adv, y, lr=0.0, apply_update=False)
    g = net.get_grads().astype(np.float64)
    d1 = net.read_activation("d1", (2560, 512))
    res[mode] = (NN.unflatten_params(g, NN.CONV_PARAM_SHAPES), d1)
    net.close()
for name, _ in NN.CONV_PARAM_SHAPES:
    a, b = res["on"][0][name], res["off"][0][name]
    d = np.abs(a - b); m = np.abs(b).max()
    print(name, "max rel", d.max() / m, "elements > 1e-6 of max:", int((d > 1e-6 * m).sum()), "of", d.size)
if res["on"][1] is not None:
    a, b = res["on"][1], res["off"][1]
    print("d1 max abs diff", np.abs(a - b).max(), "max", np.abs(b).max(), "sign flips", int(((a > 0) != (b > 0)).sum()))
