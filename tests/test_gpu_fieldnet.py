"""GPU parity of ConvPolicyVFieldNetwork (reference fed_gym/agents/paac/policy_v_network.py:83-191) against the float64 numpy
restatement oracle/nets.py:field_* with shared weights ('parity unpinned' wrt TensorFlow; the restatement's calculus is pinned
against torch autograd in tests/test_oracle_nets_torch.py).  Covers the reference's own shape test of this net
(tests/estimators_tests.py:152-215), forward and every gradient, two agents on one pixel, a clipped Adam step, bitwise
reproducibility, a second geometry, and an overfit run in the manner of the reference's train tests (:78-129)."""
import numpy as np
import pytest

from oracle import nets as NN

pytestmark = pytest.mark.gpu

REF_CONF = {'name': 'test_conv_network', 'num_actions': 3, 'clip_norm': 40., 'clip_norm_type': 'global', 'device': '/cpu:0',
            'static_size': None, 'n_agents': 10, 'entropy_regularisation_strength': 0., 'scale': 1., 'height': 32, 'width': 32,
            'channels': 3, 'filters': 5, 'conv_layers': 2}


def _bind(conf, max_samples=64):
    from goldsrl import _ffi
    from goldsrl.agents.paac.policy_v_network import ConvPolicyVFieldNetwork
    eng = _ffi.Engine(_ffi.ENV_SOLOW, 4, seed=1)       # any handle: the net only needs its device and stream
    est = ConvPolicyVFieldNetwork(conf).bind(eng, max_samples=max_samples)
    return eng, est


def _geom(conf):
    return dict(height=conf['height'], width=conf['width'], channels=conf['channels'], filters=conf['filters'],
                conv_layers=conf['conv_layers'], num_actions=conf['num_actions'])


def _shared_params(est, conf, seed=2):
    from goldsrl import _ffi_field
    shapes = NN.field_param_shapes(**_geom(conf))
    assert [tuple(s) for _, s in _ffi_field.field_param_shapes(**_geom(conf))] == [tuple(s) for _, s in shapes]
    rng = np.random.RandomState(seed)
    flat = _ffi_field.glorot_uniform_flat(3, **_geom(conf)).astype(np.float64)
    p = NN.unflatten_params(flat, shapes)
    for k in p:
        if k.endswith("_b"):
            p[k] = rng.normal(size=p[k].shape) * 0.05
    flat = NN.flatten_params(p, shapes).astype(np.float32)
    est.set_flat_params(flat)
    assert est.net.num_params == flat.size
    return NN.unflatten_params(flat.astype(np.float64), shapes), shapes, flat


def test_reference_shape_test_of_the_field_network():
    # tests/estimators_tests.py:152-215
    eng, est = _bind(REF_CONF)
    n_agents, A = 10, 3
    actions = np.random.uniform(size=(n_agents, A))
    state_idxs = np.hstack([np.random.randint(0, est.height, n_agents)[:, None], np.random.randint(0, est.width, n_agents)[:, None]])
    state = np.random.uniform(0., 1., (n_agents, est.height, est.width, est.channels))
    history = np.random.uniform(0., 1., (n_agents, 5, est.height, est.width, est.channels))
    pred = est.predict(state, history, state_idxs)
    assert pred['mu'].shape == (n_agents, A) and pred['sigma'].shape == (n_agents, A) and pred['vs'].shape == (n_agents,)
    st = est.train(state, state_idxs, actions, np.ones(n_agents), np.zeros(n_agents), lr=0.0, apply_update=False)
    assert np.isfinite(list(st.values())).all()
    assert est.net.num_params == 2 * 6144 * 3072 + 2 * 3072 + sum(int(np.prod(s)) for n, s in NN.field_param_shapes() if not n.startswith(("mu", "sigma")))
    est.net.close(); eng.close()


@pytest.mark.parametrize("conf_delta", [{}, {'height': 16, 'width': 24, 'channels': 2, 'filters': 7, 'conv_layers': 1, 'num_actions': 2,
                                             'scale': 10., 'entropy_regularisation_strength': 0.02},
                                        {'height': 16, 'width': 16, 'channels': 1, 'filters': 4, 'conv_layers': 3, 'num_actions': 1}])
def test_field_forward_and_gradients_match_oracle(conf_delta):
    conf = dict(REF_CONF, **conf_delta)
    eng, est = _bind(conf)
    p, shapes, flat = _shared_params(est, conf)
    H, W, C, A, L = conf['height'], conf['width'], conf['channels'], conf['num_actions'], conf['conv_layers']
    rng = np.random.RandomState(5)
    N = 23
    states = rng.uniform(size=(N, H, W, C)).astype(np.float32)
    pos = np.stack([rng.randint(0, H, N), rng.randint(0, W, N)], axis=1).astype(np.int32)
    pos[7] = pos[2]; pos[11] = pos[2]                   # three agents on one pixel: their head gradients add into the same columns
    pos[0] = (0, 0); pos[1] = (H - 1, W - 1)
    act = rng.uniform(size=(N, A)).astype(np.float32)
    adv, y = rng.normal(size=N).astype(np.float32), rng.normal(size=N).astype(np.float32)
    out = est.predict(states, None, pos)
    s64 = states.astype(np.float64)
    mu, sigma, vs = NN.field_forward(p, s64, pos, conf['scale'], L)
    np.testing.assert_allclose(out["mu"], mu, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["sigma"], sigma, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["vs"], vs, rtol=2e-5, atol=2e-5)
    st = est.train(states, pos, act, adv, y, lr=0.0, apply_update=False)
    loss, pl, cl, g, _ = NN.field_loss_and_grads(p, s64, pos, act.astype(np.float64), adv.astype(np.float64), y.astype(np.float64),
                                                 conf['entropy_regularisation_strength'], conf['scale'], L)
    np.testing.assert_allclose([st["loss"], st["policy_loss"], st["critic_loss_mean"]], [loss, pl, cl], rtol=1e-4, atol=1e-6)
    got = NN.unflatten_params(est.net.get_grads().astype(np.float64), shapes)
    for name, _ in shapes:
        err = np.abs(got[name] - g[name]).max() / (np.abs(g[name]).max() + 1e-12)
        assert err < 2e-4, (name, err)
    gf = NN.flatten_params(g, shapes)
    np.testing.assert_allclose(st["global_norm"], np.sqrt((gf ** 2).sum()), rtol=1e-4)
    first = est.net.get_grads()
    est.train(states, pos, act, adv, y, lr=0.0, apply_update=False)
    assert np.array_equal(first, est.net.get_grads())              # fixed-order sums: bitwise reproducible
    # one Adam step with the global-norm clip active (actor_learner.py:31-68)
    from goldsrl import _ffi_field
    net2 = _ffi_field.FieldNet(eng, max_samples=64, scale=conf['scale'], entropy_beta=conf['entropy_regularisation_strength'],
                               clip_norm=0.5 * float(st["global_norm"]), **_geom(conf))
    net2.set_params(flat)
    net2.train(states, pos, act, adv, y, lr=1e-3)
    clipped, _ = NN.clip_by_global_norm(gf, 0.5 * float(st["global_norm"]))
    ref, _, _ = NN.adam_step(flat.astype(np.float64), clipped, np.zeros_like(gf), np.zeros_like(gf), 1, 1e-3)
    assert np.abs(net2.get_params() - ref).max() <= 0.05 * 1e-3
    with pytest.raises(Exception, match="outside"):
        est.predict(states[:1], None, np.array([[H, 0]], np.int32))          # tf.gather_nd raises on an index outside the field
    net2.close(); est.net.close(); eng.close()


def test_field_network_overfits_a_fixed_batch():
    # in the manner of the reference's train tests (estimators_tests.py:78-129): repeated updates on one batch drive the loss down
    conf = dict(REF_CONF, scale=1.0)
    eng, est = _bind(conf)
    rng = np.random.RandomState(1692)
    N, A = 10, 3
    states = rng.uniform(size=(N, 32, 32, 3)).astype(np.float32)
    pos = np.stack([rng.randint(0, 32, N), rng.randint(0, 32, N)], axis=1).astype(np.int32)
    act = rng.uniform(size=(N, A)).astype(np.float32)
    adv, y = np.zeros(N, np.float32), -rng.uniform(size=N).astype(np.float32)       # critic only: vs -> y
    first = est.train(states, pos, act, adv, y, lr=1e-2)
    for _ in range(150):
        last = est.train(states, pos, act, adv, y, lr=1e-2)
    assert last["critic_loss_mean"] < 0.05 * first["critic_loss_mean"] and np.isfinite(last["loss"])
    est.net.close(); eng.close()
