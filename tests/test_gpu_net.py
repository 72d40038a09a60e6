"""GPU parity of ConvSingleAgentPolicyNetwork (HIP, fp32 MFMA) against the float64 numpy restatement
oracle/nets.py with shared weights.  Network numerics are 'parity unpinned' wrt TensorFlow (absent);
tolerance: float32 forward within 2e-5 relative of the float64 oracle, gradients within 1e-4."""
import numpy as np
import pytest

from oracle import nets as NN
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _setup(E, seed=0, biased=True):
    from goldsrl import _ffi, _ffi_net
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=11)
    eng.reset()
    rng = np.random.RandomState(seed)
    for _ in range(3):
        eng.step(O.swarm_transform_actions(rng.normal(size=(E, 10, 2)).astype(np.float32)))
    net = _ffi_net.ConvNet(eng, max_chunk_samples=40)
    flat = _ffi_net.glorot_uniform_flat(seed=3).astype(np.float64)
    p = NN.unflatten_params(flat)
    if biased:
        for k in p:
            if k.endswith("_b"):
                p[k] = rng.normal(size=p[k].shape) * 0.05
    flat = NN.flatten_params(p).astype(np.float32)
    net.set_params(flat)
    p = NN.unflatten_params(flat.astype(np.float64))
    lb, ab, pos = eng.read("locust_bins"), eng.read("agent_bins"), eng.read("positions")
    states = []
    for e in range(E):
        l = lb[e].astype(int); a = ab[e].astype(int)
        l[l[:, 0] == 255] = -1; a[a[:, 0] == 255] = -1
        states.append(O.swarm_local_states(O.swarm_grid_from_compact(l, a, 84), pos[e]))
    states = np.concatenate(states).astype(np.float32).astype(np.float64)    # the TF placeholder is float32
    return eng, net, p, states, (lb, ab, pos)


def test_param_vector_layout():
    from goldsrl import _ffi_net
    assert [tuple(x) for x in _ffi_net.CONV_PARAM_SHAPES] == [tuple(x) for x in NN.CONV_PARAM_SHAPES]
    assert _ffi_net.glorot_uniform_flat().size == 2210213


def test_conv_forward_matches_oracle_layer_by_layer():
    E = 6    # 60 samples: two chunks of 40 -> exercises chunking and partial tiles
    eng, net, p, states, obs = _setup(E)
    out = net.predict()
    mu, sigma, vs, c = NN.conv_forward(p, states, 1000.0, keep=True)
    # activations of the LAST chunk (envs 4,5 -> samples 40..59)
    n_last = 20
    for name, ref in (("a1", c["a1"]), ("a2", c["a2"]), ("a3", c["a3"]), ("d1", c["d1"]), ("d2", c["d2"]), ("p1", c["p1"]),
                      ("v1", c["v1"]), ("v2", c["v2"])):
        ref = ref[40:]
        got = net.read_activation(name, ref.shape)
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6, err_msg=name)
    np.testing.assert_allclose(out["mu"], mu, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["sigma"], sigma, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["vs"], vs, rtol=2e-5, atol=1e-3)
    assert out["mu"].shape == (60, 2) and out["vs"].shape == (60,)          # estimators_tests.py:72-76
    # predict on caller-supplied observations gives the same numbers
    out2 = net.predict_obs(*obs)
    assert np.array_equal(out2["mu"], out["mu"]) and np.array_equal(out2["vs"], out["vs"])


def test_conv_forward_zero_bias_default_init():
    eng, net, p, states, obs = _setup(4, biased=False)
    out = net.predict()
    mu, sigma, vs = NN.conv_forward(p, states, 1000.0)
    np.testing.assert_allclose(out["mu"], mu, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["vs"], vs, rtol=2e-5, atol=1e-3)
