"""The network oracle (oracle/nets.py) is 'parity unpinned' (no TF here): check it against itself --
analytic gradients vs central finite differences, parameter count vs SURVEY N1, GRU masking."""
import numpy as np

from oracle import nets as NN


def _tiny_batch(n=3, seed=0):
    rng = np.random.RandomState(seed)
    states = np.zeros((n, 84, 84, 3))
    for i in range(n):
        for _ in range(80):
            states[i, rng.randint(84), rng.randint(84), 0] += 1 / 80.0
        for _ in range(10):
            states[i, rng.randint(84), rng.randint(84), 1] += 1 / 10.0
        states[i, rng.randint(84), rng.randint(84), 2] = 1.0
    return states, rng.normal(size=(n, 2)), rng.normal(size=n) * 0.01, -rng.rand(n) * 300


def test_param_count_matches_survey():
    assert NN.CONV_NUM_PARAMS == 2210213
    p = NN.conv_init()
    flat = NN.flatten_params(p)
    assert flat.size == 2210213
    q = NN.unflatten_params(flat)
    assert all(np.array_equal(p[k], q[k]) for k in p)
    n_flat = sum(int(np.prod(s)) for _, s in NN.flat_param_shapes())
    assert 30000 < n_flat < 32000          # "~30.7k params" 


def test_conv_shapes_and_ranges():
    p = NN.conv_init()
    s, a, adv, y = _tiny_batch()
    mu, sigma, vs = NN.conv_forward(p, s, 1000.0)
    assert mu.shape == (3, 2) and sigma.shape == (3, 2) and vs.shape == (3,)     # estimators_tests.py:72-76
    assert (np.abs(mu) <= 1).all() and (sigma > 0).all() and (sigma < 1).all() and (vs < 0).all()


def test_conv_gradients_match_finite_differences():
    p = NN.conv_init(seed=5)
    for k in p:                      # move biases off zero so ReLU gates are generic
        if k.endswith("_b"):
            p[k] = np.random.RandomState(1).normal(size=p[k].shape) * 0.05
    s, a, adv, y = _tiny_batch()
    beta, scale = 0.02, 1000.0
    loss, pl, cl, g, _ = NN.conv_loss_and_grads(p, s, a, adv, y, beta, scale)
    f = lambda q: NN.conv_loss_and_grads(q, s, a, adv, y, beta, scale)[0]
    num = NN.numeric_grad(f, p, [n for n, _ in NN.CONV_PARAM_SHAPES], eps=1e-6, max_per=3)
    for n, vals in num.items():
        for idx, gv in vals:
            assert abs(g[n][idx] - gv) <= 3e-3 * abs(gv) + 2e-6   # ReLU kinks make FD noisy at the 1e-3 level, (n, idx, g[n][idx], gv)


def test_clip_and_adam():
    g = np.array([3.0, 4.0])
    c, n = NN.clip_by_global_norm(g, 1.0)
    assert n == 5.0 and np.allclose(c, g / 5.0)
    c, n = NN.clip_by_global_norm(g, 40.0)
    assert np.array_equal(c, g)
    p, m, v = NN.adam_step(np.zeros(2), g, np.zeros(2), np.zeros(2), 1, 1e-3)
    assert np.allclose(p, -1e-3 * np.sign(g), rtol=1e-6)      # first Adam step is lr*sign(g)


def test_gru_sequence_length_masking():
    p = NN.flat_init()
    rng = np.random.RandomState(0)
    hist = rng.normal(size=(4, 5, 2))
    hist[1, 2:] = 0; hist[2, 1:] = 0
    h = NN.gru_last_state(p, hist)
    h_short = NN.gru_last_state(p, hist[1:2, :2])
    assert np.allclose(h[1], h_short[0])
    mu, sigma, vs = NN.flat_forward(p, rng.normal(size=(4, 2)), hist, 100.0)
    assert (np.abs(mu) <= 5).all() and (sigma > 1e-3).all() and mu.shape == (4, 1)


def test_flat_gradients_match_finite_differences():
    p = NN.flat_init(seed=4)
    rng = np.random.RandomState(2)
    for k in p:
        if k.endswith("_b"):
            p[k] = p[k] + rng.normal(size=p[k].shape) * 0.1
    n = 6
    states = rng.normal(size=(n, 2)) * [0.3, 0.1] + [0.65, 0.0]
    hist = np.repeat(states[:, None], 5, 1)
    for i, L in enumerate((1, 2, 3, 5, 5, 4)):
        hist[i, L:] = 0                                   # quirk Q11 windows of different lengths
    act, adv, y = rng.normal(size=(n, 1)), rng.normal(size=n) * 0.5, rng.normal(size=n) * 50
    loss, pl, cl, g, (mu, sigma, vs) = NN.flat_loss_and_grads(p, states, hist, act, adv, y, 100.0)
    l2, pl2, cl2 = NN.flat_loss(p, states, hist, act, adv, y, 100.0)
    assert abs(loss - l2) < 1e-12
    shapes = NN.flat_param_shapes()
    f = lambda q: NN.flat_loss(q, states, hist, act, adv, y, 100.0)[0]
    num = NN.numeric_grad(f, p, [nm for nm, _ in shapes], eps=1e-6, max_per=4)
    for nm, vals in num.items():
        for idx, gv in vals:
            assert abs(g[nm][idx] - gv) <= 2e-3 * abs(gv) + 1e-6, (nm, idx, g[nm][idx], gv)


def test_the_relu_flip_explainer_takes_a_flip_whole_or_not_at_all():
    """oracle.nets.explain_by_relu_flips (what tests/test_gpu_net_tiles.py holds the device's gradient against at 640-1 920 samples)
    must explain a REAL flipped ReLU derivative -- and nothing else: a gradient that contains 0.4 of a candidate's effect, or a
    candidate's effect plus a genuine error, keeps its residual (advisor r4: the free least-squares fit it
    replaces could absorb such things)."""
    p = NN.conv_init(seed=5)
    rs = np.random.RandomState(1)
    for k in p:
        if k.endswith("_b"):
            p[k] = rs.normal(size=p[k].shape) * 0.05
    s, a, adv, y = _tiny_batch(n=6, seed=3)
    beta, scale = 0.02, 1000.0
    eps = {k: 2e-3 for k in NN.RELU_LAYERS}
    eps["a1"] = 0.0      # (conv1 has ~10^4 elements per sample within any useful eps of zero: the dense layers give enough candidates)
    eps["a2"] = eps["a3"] = 2e-4
    amb = NN.relu_ambiguous(p, s, eps)
    assert 3 <= len(amb) < 400, len(amb)
    _, _, _, g, _ = NN.conv_loss_and_grads(p, s, a, adv, y, beta, scale)
    ref_flat = NN.flatten_params(g)
    # a candidate with a visible effect
    sizes = {"a1": 20 * 20 * 32, "a2": 9 * 9 * 64, "a3": 7 * 7 * 64, "d1": 512, "d2": 256, "p1": 512, "v1": 512, "v2": 256}
    pick = None
    for j, (smp, layer, idx, _) in enumerate(amb):
        gf = NN.conv_loss_and_grads(p, s, a, adv, y, beta, scale, flip=[(layer, smp * sizes[layer] + idx)])[3]
        d = NN.flatten_params(gf) - ref_flat
        rel = max(np.abs(NN.unflatten_params(d)[k]).max() / np.abs(g[k]).max() for k in g)
        if rel > 1e-3:
            pick = (j, d)
            break
    assert pick is not None
    j, d = pick
    noise = 1.0 + 1e-7 * rs.normal(size=ref_flat.size)

    def worst(res):
        return max(np.abs(res[k]).max() / np.abs(g[k]).max() for k in res)
    # (1) the flip as it happens: taken whole, residual at float32 level
    got = ((ref_flat + d) * noise).astype(np.float32)
    res, used, amb2 = NN.explain_by_relu_flips(p, s, a, adv, y, beta, scale, got, g, eps=eps)
    assert len(amb2) == len(amb) and used[j] == 1.0 and used.sum() <= 3 and worst(res) < 2e-5, (used.nonzero(), worst(res))
    # (2) no flip: nothing taken
    res, used, _ = NN.explain_by_relu_flips(p, s, a, adv, y, beta, scale, (ref_flat * noise).astype(np.float32), g, eps=eps)
    assert used.sum() == 0 and worst(res) < 2e-5
    # (3) 0.4 of a flip's effect is an error, not a flip: the residual stays
    res, used, _ = NN.explain_by_relu_flips(p, s, a, adv, y, beta, scale, (ref_flat + 0.4 * d).astype(np.float32), g, eps=eps)
    assert worst(res) > 1e-4, (used.nonzero(), worst(res))
    # (4) a flip plus a genuine error in one block: the flip is explained, the error is not
    e = np.zeros_like(ref_flat)
    off = 0
    for name, shape in NN.CONV_PARAM_SHAPES:
        sz = int(np.prod(shape))
        if name == "dense2_w":
            e[off:off + sz] = 1e-3 * np.abs(g[name]).max() * rs.normal(size=sz)
        off += sz
    res, used, _ = NN.explain_by_relu_flips(p, s, a, adv, y, beta, scale, (ref_flat + d + e).astype(np.float32), g, eps=eps)
    left = {k: np.abs(res[k]).max() / np.abs(g[k]).max() for k in res}
    assert used[j] == 1.0 and left["dense2_w"] > 5e-4, (used.nonzero(), left)
