"""oracle/nets.py (hand-derived numpy forward/backward of N1 and N2) against an INDEPENDENT implementation: the same
networks written with torch ops (float64, CPU) and differentiated by autograd.  TensorFlow is not available to pin the
reference's numerics ("parity unpinned"); this at least pins the restatement's calculus: forward
values and every parameter gradient agree to 1e-9.  TF conventions mapped by hand: NHWC inputs, [kh,kw,cin,cout]
kernels, VALID padding, tf.layers.flatten order (h,w,c), GRUCell gate order (r,u) with the candidate on [x, r*h]."""
import numpy as np
import pytest

from oracle import nets as NN

torch = pytest.importorskip("torch")

LOG_2PI = float(np.log(2 * np.pi))


def _t(p):
    return {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}


def _conv(x, w, b, stride):      # x NHWC, w HWIO
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), bias=b, stride=stride)
    return y.permute(0, 2, 3, 1)


def _gauss_loss(mu, sigma, actions, adv, y, vs, beta, scale, entropy_in_loss):
    diff = actions - mu
    logp = -0.5 * (diff / sigma) ** 2 - torch.log(sigma) - 0.5 * LOG_2PI
    ent = 0.5 + 0.5 * LOG_2PI + torch.log(sigma)
    if entropy_in_loss:      # policy_v_network.py:45-56
        pl = -torch.mean(logp.sum(1) * adv + beta * ent.sum(1))
    else:                    # :228-235
        pl = torch.mean(-logp * adv[:, None])
    cl = torch.mean(0.25 * (vs - y) ** 2 / scale)
    return pl + cl, pl, cl


def test_conv_net_matches_torch_autograd():
    rng = np.random.RandomState(0)
    p = NN.unflatten_params(np.concatenate([rng.normal(size=int(np.prod(s))) * (0.05 if n.endswith("_w") else 0.02)
                                            for n, s in NN.CONV_PARAM_SHAPES]))
    N = 3
    states = (rng.uniform(size=(N, 84, 84, 3)) < 0.02).astype(np.float64) * rng.uniform(0.1, 1.0, size=(N, 84, 84, 3))
    actions, adv, y = rng.normal(size=(N, 2)), rng.normal(size=N) * 0.01, rng.normal(size=N) * 50.0
    beta, scale = 0.02, 1000.0
    loss, pl, cl, g, (mu, sigma, vs) = NN.conv_loss_and_grads(p, states, actions, adv, y, beta, scale)

    tp = _t(p)
    x = torch.tensor(states)
    a1 = torch.relu(_conv(x, tp["conv1_w"], tp["conv1_b"], 4))
    a2 = torch.relu(_conv(a1, tp["conv2_w"], tp["conv2_b"], 2))
    a3 = torch.relu(_conv(a2, tp["conv3_w"], tp["conv3_b"], 1))
    flat = a3.reshape(N, -1)
    d1 = torch.relu(flat @ tp["dense1_w"] + tp["dense1_b"])
    d2 = torch.relu(d1 @ tp["dense2_w"] + tp["dense2_b"])
    p1 = torch.relu(d2 @ tp["pol1_w"] + tp["pol1_b"])
    tmu = torch.tanh(p1 @ tp["mu_w"] + tp["mu_b"])
    tsg = torch.sigmoid(p1 @ tp["sigma_w"] + tp["sigma_b"])
    v1 = torch.relu(d2 @ tp["v1_w"] + tp["v1_b"])
    v2 = torch.relu(v1 @ tp["v2_w"] + tp["v2_b"])
    tvs = -scale * torch.nn.functional.softplus((v2 @ tp["v3_w"] + tp["v3_b"])[:, 0])
    tl, tpl, tcl = _gauss_loss(tmu, tsg, torch.tensor(actions), torch.tensor(adv), torch.tensor(y), tvs, beta, scale, True)
    tl.backward()

    np.testing.assert_allclose(mu, tmu.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sigma, tsg.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(vs, tvs.detach().numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose([loss, pl, cl], [tl.item(), tpl.item(), tcl.item()], rtol=1e-10)
    for name, _ in NN.CONV_PARAM_SHAPES:
        ref = tp[name].grad.numpy()
        np.testing.assert_allclose(g[name], ref, rtol=1e-8, atol=1e-10 * (np.abs(ref).max() + 1e-30), err_msg=name)


@pytest.mark.parametrize("dims", [dict(static_size=2, temporal_size=2, num_actions=1), dict(static_size=5, temporal_size=5, num_actions=2)])
def test_flat_net_matches_torch_autograd(dims):
    rng = np.random.RandomState(1)
    p = NN.flat_init(seed=4, **dims)
    for k in p:      # non-trivial biases everywhere
        if k.endswith("_b"):
            p[k] = p[k] + rng.normal(size=p[k].shape) * 0.1
    N, T, D, S0, A, H = 7, 5, dims["temporal_size"], dims["static_size"], dims["num_actions"], 32
    states = rng.normal(size=(N, S0))
    hist = rng.normal(size=(N, T, D))
    hist[0, 2:] = 0.0      # sequence_length 2
    hist[1, 4:] = 0.0      # 4
    hist[2, 1:] = 0.0      # 1
    actions, adv, y = rng.normal(size=(N, A)), rng.normal(size=N) * 0.1, rng.normal(size=N) * 5.0
    scale, ub, lb = 100.0, 5.0, -5.0
    loss, pl, cl, g, (mu, sigma, vs) = NN.flat_loss_and_grads(p, states, hist, actions, adv, y, scale)

    tp = _t(p)
    x_s, x_h = torch.tensor(states), torch.tensor(hist)
    length = torch.tensor(np.sign(np.max(np.abs(hist), axis=2)).sum(axis=1).astype(int))
    h = torch.zeros((N, H), dtype=torch.float64)
    for t in range(T):
        xt = x_h[:, t]
        gates = torch.sigmoid(torch.cat([xt, h], 1) @ tp["gru_gates_w"] + tp["gru_gates_b"])
        r, u = gates[:, :H], gates[:, H:]
        c = torch.tanh(torch.cat([xt, r * h], 1) @ tp["gru_cand_w"] + tp["gru_cand_b"])
        h = torch.where((t < length)[:, None], u * h + (1 - u) * c, h)
    dt_ = torch.relu(h @ tp["temporal_w"] + tp["temporal_b"])
    s1 = torch.relu(x_s @ tp["static1_w"] + tp["static1_b"])
    s2 = torch.relu(s1 @ tp["static2_w"] + tp["static2_b"])
    x96 = torch.cat([dt_, s2], 1)
    m = torch.tanh(torch.relu(x96 @ tp["mu1_w"] + tp["mu1_b"]) @ tp["mu2_w"] + tp["mu2_b"])
    tmu = ((ub - lb) / 2.0) * torch.tanh(m @ tp["mu3_w"] + tp["mu3_b"]) + (lb + ub) / 2.0
    s = torch.tanh(torch.relu(x96 @ tp["sig1_w"] + tp["sig1_b"]) @ tp["sig2_w"] + tp["sig2_b"])
    tsg = torch.sigmoid(s @ tp["sig3_w"] + tp["sig3_b"]) + 1e-3
    tvs = scale * (torch.tanh(x96 @ tp["v1_w"] + tp["v1_b"]) @ tp["v2_w"] + tp["v2_b"])[:, 0]
    tl, tpl, tcl = _gauss_loss(tmu, tsg, torch.tensor(actions), torch.tensor(adv), torch.tensor(y), tvs, 0.0, scale, False)
    tl.backward()

    np.testing.assert_allclose(mu, tmu.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sigma, tsg.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(vs, tvs.detach().numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose([loss, pl, cl], [tl.item(), tpl.item(), tcl.item()], rtol=1e-10)
    for name, _ in NN.flat_param_shapes(**dims):
        ref = tp[name].grad.numpy()
        np.testing.assert_allclose(g[name], ref, rtol=1e-8, atol=1e-10 * (np.abs(ref).max() + 1e-30), err_msg=name)


def test_field_net_matches_torch_autograd():
    """ConvPolicyVFieldNetwork (policy_v_network.py:83-191): 3x3 'same' convs + 2x2 max-pools, dense stack, Dense(H*W*A) heads
    gathered at the agent's position -- the numpy restatement's forward and every gradient against torch autograd, at the
    reference test's geometry (tests/estimators_tests.py:152-176) incl. two agents on the same pixel."""
    H = W = 32
    C, F, L, A, N = 3, 5, 2, 3, 6
    shapes = NN.field_param_shapes(H, W, C, F, L, A)
    rng = np.random.RandomState(1)
    p = {n: rng.normal(size=s) * (0.3 / np.sqrt(s[0] if len(s) == 2 else 27) if n.endswith("_w") else 0.05) for n, s in shapes}
    states = rng.uniform(size=(N, H, W, C))
    positions = np.stack([rng.randint(0, H, N), rng.randint(0, W, N)], axis=1)
    positions[3] = positions[1]                      # shared output pixel: gradients of both samples add
    actions, adv, y = rng.uniform(size=(N, A)), rng.normal(size=N), rng.normal(size=N)
    beta, scale = 0.02, 10.0
    loss, pl, cl, g, (mu, sigma, vs) = NN.field_loss_and_grads(p, states, positions, actions, adv, y, beta, scale, L)
    assert mu.shape == (N, A) and sigma.shape == (N, A) and vs.shape == (N,)

    tp = _t(p)
    x = torch.tensor(states)
    for i in range(L):
        z = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), tp["conv%d_w" % i].permute(3, 2, 0, 1), bias=tp["conv%d_b" % i], padding=1)
        x = torch.nn.functional.max_pool2d(torch.relu(z), 2).permute(0, 2, 3, 1)
    flat = x.reshape(N, -1)
    d1 = torch.relu(flat @ tp["dense1_w"] + tp["dense1_b"])
    d2 = torch.relu(d1 @ tp["dense2_w"] + tp["dense2_b"])
    p1 = torch.relu(d2 @ tp["pol1_w"] + tp["pol1_b"])
    p2 = torch.relu(p1 @ tp["pol2_w"] + tp["pol2_b"])
    mus = torch.tanh(p2 @ tp["mu_w"] + tp["mu_b"]).reshape(N, H, W, A)             # the full field, as the reference builds it
    sgs = torch.sigmoid(p2 @ tp["sigma_w"] + tp["sigma_b"]).reshape(N, H, W, A)
    idx = torch.arange(N)
    tmu, tsg = mus[idx, torch.tensor(positions[:, 0]), torch.tensor(positions[:, 1])], sgs[idx, torch.tensor(positions[:, 0]), torch.tensor(positions[:, 1])]
    v1 = torch.relu(d2 @ tp["v1_w"] + tp["v1_b"])
    v2 = torch.relu(v1 @ tp["v2_w"] + tp["v2_b"])
    tvs = -scale * torch.nn.functional.softplus((v2 @ tp["v3_w"] + tp["v3_b"])[:, 0])
    tl, tpl, tcl = _gauss_loss(tmu, tsg, torch.tensor(actions), torch.tensor(adv), torch.tensor(y), tvs, beta, scale, True)
    tl.backward()
    np.testing.assert_allclose(mu, tmu.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sigma, tsg.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(vs, tvs.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose([loss, pl, cl], [tl.item(), tpl.item(), tcl.item()], rtol=1e-10)
    for name, _ in shapes:
        np.testing.assert_allclose(g[name], tp[name].grad.numpy(), rtol=1e-8, atol=1e-11, err_msg=name)
