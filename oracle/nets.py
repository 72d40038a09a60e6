"""CPU ORACLE for the policy/value networks -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference builds these nets from tensorflow==1.4.1 layers
(/root/reference/requirements.txt:33), which is neither under /root/reference nor installable
here, and the reference's own tests pin only output SHAPES and overfit-to-zero behaviour
(/root/reference/tests/estimators_tests.py:24-129).  This file restates, in float64 numpy, the
graph that /root/reference/fed_gym/agents/paac/policy_v_network.py:5-66 (ConvSingleAgentPolicyNetwork)
and :194-251 (FlatPolicyVNetwork, with a3c/estimators.py:5-28) describe, using the published
TF-1.4 semantics of tf.layers.Conv2D/Dense (VALID padding, NHWC, kernel [kh,kw,cin,cout], glorot
uniform, zero bias), tf.nn.rnn_cell.GRUCell, tf.distributions.Normal, tf.clip_by_global_norm and
tf.train.AdamOptimizer (/root/reference/fed_gym/agents/paac/actor_learner.py:31-68).
The HIP kernels are checked against THIS restatement with shared weights, and this restatement is
checked against itself by finite differences (tests/test_oracle_nets.py).
"""
import numpy as np

LOG_2PI = np.log(2.0 * np.pi)

# (name, shape) in the order tf.trainable_variables() creates them (policy_v_network.py:14-59)
CONV_PARAM_SHAPES = [
    ("conv1_w", (8, 8, 3, 32)), ("conv1_b", (32,)),
    ("conv2_w", (4, 4, 32, 64)), ("conv2_b", (64,)),
    ("conv3_w", (3, 3, 64, 64)), ("conv3_b", (64,)),
    ("dense1_w", (3136, 512)), ("dense1_b", (512,)),
    ("dense2_w", (512, 256)), ("dense2_b", (256,)),
    ("pol1_w", (256, 512)), ("pol1_b", (512,)),
    ("mu_w", (512, 2)), ("mu_b", (2,)),
    ("sigma_w", (512, 2)), ("sigma_b", (2,)),
    ("v1_w", (256, 512)), ("v1_b", (512,)),
    ("v2_w", (512, 256)), ("v2_b", (256,)),
    ("v3_w", (256, 1)), ("v3_b", (1,)),
]
CONV_NUM_PARAMS = sum(int(np.prod(s)) for _, s in CONV_PARAM_SHAPES)   # 2 210 213


def glorot_uniform(rng, shape):
    """tf.glorot_uniform_initializer: U(-l, l), l = sqrt(6/(fan_in+fan_out)); for conv kernels
    fan_in = kh*kw*cin, fan_out = kh*kw*cout."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def conv_param_shapes(num_actions=2):
    """CONV_PARAM_SHAPES with Dense(num_actions) mu / sigma heads (policy_v_network.py:40-43)."""
    return [(n, (s[0], num_actions) if n in ("mu_w", "sigma_w") else ((num_actions,) if n in ("mu_b", "sigma_b") else s)) for n, s in CONV_PARAM_SHAPES]


def conv_init(seed=3):
    rng = np.random.RandomState(seed)
    return {n: (glorot_uniform(rng, s) if n.endswith("_w") else np.zeros(s)) for n, s in CONV_PARAM_SHAPES}


def flatten_params(p, shapes=CONV_PARAM_SHAPES):
    return np.concatenate([np.asarray(p[n], dtype=np.float64).reshape(-1) for n, _ in shapes])


def unflatten_params(flat, shapes=CONV_PARAM_SHAPES):
    out, o = {}, 0
    for n, s in shapes:
        k = int(np.prod(s))
        out[n] = np.asarray(flat[o:o + k]).reshape(s)
        o += k
    return out


def _im2col(x, kh, kw, stride):
    """x (N,H,W,C) -> (N,OH,OW,kh*kw*C) with the (ky,kx,c) ordering of a TF [kh,kw,cin,cout] kernel."""
    N, H, W, C = x.shape
    oh, ow = (H - kh) // stride + 1, (W - kw) // stride + 1
    cols = np.empty((N, oh, ow, kh, kw, C), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            cols[:, :, :, ky, kx, :] = x[:, ky:ky + stride * oh:stride, kx:kx + stride * ow:stride, :]
    return cols.reshape(N, oh, ow, kh * kw * C)


def _col2im(dcols, xshape, kh, kw, stride):
    N, H, W, C = xshape
    oh, ow = (H - kh) // stride + 1, (W - kw) // stride + 1
    d = dcols.reshape(N, oh, ow, kh, kw, C)
    dx = np.zeros(xshape, dtype=dcols.dtype)
    for ky in range(kh):
        for kx in range(kw):
            dx[:, ky:ky + stride * oh:stride, kx:kx + stride * ow:stride, :] += d[:, :, :, ky, kx, :]
    return dx


def _conv(x, w, b, stride):
    kh, kw, cin, cout = w.shape
    cols = _im2col(x, kh, kw, stride)
    return cols @ w.reshape(-1, cout) + b, cols


def softplus(x):
    return np.logaddexp(0.0, x)


def _sigmoid(x):
    return 0.5 * (1.0 + np.tanh(0.5 * x))


def conv_forward(p, states, scale, keep=False):
    """ConvSingleAgentPolicyNetwork forward (policy_v_network.py:14-59).
    states (N,84,84,3).  Returns mu (N,2), sigma (N,2), vs (N,) [+ cache]."""
    c = {}
    z1, c["cols1"] = _conv(states, p["conv1_w"], p["conv1_b"], 4); a1 = np.maximum(z1, 0)
    z2, c["cols2"] = _conv(a1, p["conv2_w"], p["conv2_b"], 2); a2 = np.maximum(z2, 0)
    z3, c["cols3"] = _conv(a2, p["conv3_w"], p["conv3_b"], 1); a3 = np.maximum(z3, 0)
    flat = a3.reshape(a3.shape[0], -1)                       # tf.layers.flatten of NHWC: (h, w, c)
    d1 = np.maximum(flat @ p["dense1_w"] + p["dense1_b"], 0)
    d2 = np.maximum(d1 @ p["dense2_w"] + p["dense2_b"], 0)   # processed_state
    p1 = np.maximum(d2 @ p["pol1_w"] + p["pol1_b"], 0)
    mu = np.tanh(p1 @ p["mu_w"] + p["mu_b"])
    sigma = _sigmoid(p1 @ p["sigma_w"] + p["sigma_b"])
    v1 = np.maximum(d2 @ p["v1_w"] + p["v1_b"], 0)
    v2 = np.maximum(v1 @ p["v2_w"] + p["v2_b"], 0)
    zv = (v2 @ p["v3_w"] + p["v3_b"])[:, 0]
    vs = -scale * softplus(zv)                               # policy_v_network.py:59
    if keep:
        c.update(a1=a1, a2=a2, a3=a3, flat=flat, d1=d1, d2=d2, p1=p1, mu=mu, sigma=sigma, v1=v1, v2=v2, zv=zv, vs=vs,
                 states_shape=states.shape)
        return mu, sigma, vs, c
    return mu, sigma, vs


def gaussian_loss_terms(mu, sigma, actions, advantages, critic_target, vs, beta, scale, entropy_in_loss=True):
    """Loss of policy_v_network.py:45-66 (and :228-251 with entropy_in_loss=False, per-element nll).
    Returns loss, policy_loss, critic_loss_mean, and d(loss)/d(mu, sigma, vs)."""
    N, A = mu.shape
    diff = actions - mu
    logp = -0.5 * (diff / sigma) ** 2 - np.log(sigma) - 0.5 * LOG_2PI      # Normal.log_prob
    ent = 0.5 + 0.5 * LOG_2PI + np.log(sigma)                                # Normal.entropy
    if entropy_in_loss:
        # -mean_n( sum_a logp * adv + beta * sum_a ent )
        policy_loss = -np.mean(logp.sum(axis=1) * advantages + beta * ent.sum(axis=1))
        dlogp = -(advantages / N)[:, None] * np.ones_like(mu)
        dent = -(beta / N) * np.ones_like(mu)
    else:
        # mean over (n, a) of (-logp * adv[:, None]); no entropy term in the loss (:230-235)
        policy_loss = np.mean(-logp * advantages[:, None])
        dlogp = -(advantages / (N * A))[:, None] * np.ones_like(mu)
        dent = np.zeros_like(mu)
    dmu = dlogp * (diff / sigma ** 2)
    dsigma = dlogp * ((diff ** 2) / sigma ** 3 - 1.0 / sigma) + dent / sigma
    critic_loss = (vs - critic_target) ** 2 / scale
    critic_loss_mean = np.mean(0.25 * critic_loss)
    dvs = 0.5 * (vs - critic_target) / (scale * N)
    return policy_loss + critic_loss_mean, policy_loss, critic_loss_mean, dmu, dsigma, dvs


RELU_LAYERS = ("a1", "a2", "a3", "d1", "d2", "p1", "v1", "v2")


def _conv_backward(p, c, mu, sigma, dmu, dsigma, dvs, m, scale):
    """d loss / d params of N1 from the head gradients, the forward cache c and the ReLU derivative masks m."""
    g = {}
    dzmu = dmu * (1 - mu ** 2)
    dzsig = dsigma * sigma * (1 - sigma)
    g["mu_w"], g["mu_b"] = c["p1"].T @ dzmu, dzmu.sum(0)
    g["sigma_w"], g["sigma_b"] = c["p1"].T @ dzsig, dzsig.sum(0)
    dp1 = (dzmu @ p["mu_w"].T + dzsig @ p["sigma_w"].T) * m["p1"]
    g["pol1_w"], g["pol1_b"] = c["d2"].T @ dp1, dp1.sum(0)
    dzv = (dvs * (-scale) * _sigmoid(c["zv"]))[:, None]
    g["v3_w"], g["v3_b"] = c["v2"].T @ dzv, dzv.sum(0)
    dv2 = (dzv @ p["v3_w"].T) * m["v2"]
    g["v2_w"], g["v2_b"] = c["v1"].T @ dv2, dv2.sum(0)
    dv1 = (dv2 @ p["v2_w"].T) * m["v1"]
    g["v1_w"], g["v1_b"] = c["d2"].T @ dv1, dv1.sum(0)
    dd2 = (dp1 @ p["pol1_w"].T + dv1 @ p["v1_w"].T) * m["d2"]
    g["dense2_w"], g["dense2_b"] = c["d1"].T @ dd2, dd2.sum(0)
    dd1 = (dd2 @ p["dense2_w"].T) * m["d1"]
    g["dense1_w"], g["dense1_b"] = c["flat"].T @ dd1, dd1.sum(0)
    da3 = (dd1 @ p["dense1_w"].T).reshape(c["a3"].shape) * m["a3"]
    g["conv3_w"] = (c["cols3"].reshape(-1, c["cols3"].shape[-1]).T @ da3.reshape(-1, 64)).reshape(p["conv3_w"].shape)
    g["conv3_b"] = da3.sum(axis=(0, 1, 2))
    da2 = _col2im(da3 @ p["conv3_w"].reshape(-1, 64).T, c["a2"].shape, 3, 3, 1) * m["a2"]
    g["conv2_w"] = (c["cols2"].reshape(-1, c["cols2"].shape[-1]).T @ da2.reshape(-1, 64)).reshape(p["conv2_w"].shape)
    g["conv2_b"] = da2.sum(axis=(0, 1, 2))
    da1 = _col2im(da2 @ p["conv2_w"].reshape(-1, 64).T, c["a1"].shape, 4, 4, 2) * m["a1"]
    g["conv1_w"] = (c["cols1"].reshape(-1, c["cols1"].shape[-1]).T @ da1.reshape(-1, 32)).reshape(p["conv1_w"].shape)
    g["conv1_b"] = da1.sum(axis=(0, 1, 2))
    return g


def conv_loss_and_grads(p, states, actions, advantages, critic_target, beta, scale, flip=None):
    """loss and d loss / d params for N1 (float64).
    flip: optional list of (layer, flat index into that layer's activation array) whose ReLU DERIVATIVE is inverted in the backward
    pass (the forward values stay): what a float32 evaluation does when it puts a pre-activation that float64 sees within
    round-off of zero on the other side (relu_ambiguous / explain_by_relu_flips below)."""
    mu, sigma, vs, c = conv_forward(p, states, scale, keep=True)
    loss, pl, cl, dmu, dsigma, dvs = gaussian_loss_terms(mu, sigma, actions, advantages, critic_target, vs, beta, scale)
    m = {k: c[k] > 0 for k in RELU_LAYERS}
    for layer, idx in (flip or ()):
        m[layer].reshape(-1)[idx] ^= True
    return loss, pl, cl, _conv_backward(p, c, mu, sigma, dmu, dsigma, dvs, m, scale), (mu, sigma, vs)


# float32 evaluations differ from float64 by ~5e-7 of a layer's largest value in the GEMM layers (measured); conv1
# is a sum of <= 91 products per pixel
RELU_EPS = {"a1": 3e-7, "a2": 2e-6, "a3": 2e-6, "d1": 2e-6, "d2": 2e-6, "p1": 2e-6, "v1": 2e-6, "v2": 2e-6}


def relu_ambiguous(p, states, eps=None, batch=128):
    """ReLU inputs of N1 that float64 sees within eps[layer] * (the layer's largest |pre-activation|) of zero: where a float32
    evaluation may legitimately take the other branch.  Returns a list of (sample, layer, flat index inside the sample's
    activation, z)."""
    eps = RELU_EPS if eps is None else (eps if isinstance(eps, dict) else {k: eps for k in RELU_LAYERS})
    out = []
    for s0 in range(0, states.shape[0], batch):
        x = states[s0:s0 + batch]
        z = {}
        z["a1"], _ = _conv(x, p["conv1_w"], p["conv1_b"], 4); a = np.maximum(z["a1"], 0)
        z["a2"], _ = _conv(a, p["conv2_w"], p["conv2_b"], 2); a = np.maximum(z["a2"], 0)
        z["a3"], _ = _conv(a, p["conv3_w"], p["conv3_b"], 1); a = np.maximum(z["a3"], 0)
        z["d1"] = a.reshape(a.shape[0], -1) @ p["dense1_w"] + p["dense1_b"]; d1 = np.maximum(z["d1"], 0)
        z["d2"] = d1 @ p["dense2_w"] + p["dense2_b"]; d2 = np.maximum(z["d2"], 0)
        z["p1"] = d2 @ p["pol1_w"] + p["pol1_b"]
        z["v1"] = d2 @ p["v1_w"] + p["v1_b"]; v1 = np.maximum(z["v1"], 0)
        z["v2"] = v1 @ p["v2_w"] + p["v2_b"]
        for layer in RELU_LAYERS:
            zl = z[layer].reshape(x.shape[0], -1)
            near = np.argwhere(np.abs(zl) < eps[layer] * np.abs(zl).max())
            out.extend((s0 + int(i), layer, int(j), float(zl[i, j])) for i, j in near)
    return out


def explain_by_relu_flips(p, states, actions, advantages, critic_target, beta, scale, got_flat, g_ref, eps=None, max_flips=32):
    """Compares a float32 gradient (flat vector) with the float64 one up to the ReLU decisions float32 cannot be held to.
    A sample's pre-activation within round-off of zero may fall on either side in float32; its derivative mask then flips and the
    gradient moves by that element's whole contribution -- 1e-3 of a block's largest entry is typical at ~1000 samples -- although
    nothing is wrong.  For every ambiguous element j (relu_ambiguous) the exact gradient change D_j of flipping it is computed from
    the one sample it belongs to.

    A flip HAPPENED or it did not: got - ref must be the sum of D_j over a small SET of candidates, every coefficient exactly 1
    (round 5; until then a least-squares fit with free real coefficients, which could absorb part of a genuine error lying in the
    span of a few hundred columns).  The set is built greedily -- the candidate whose whole effect takes most off the residual
    (every block weighted by its own largest reference entry), as long as taking it removes at least half of its own squared norm,
    i.e. the residual really contains that column once and not 0.4 of it -- and nothing is ever fitted.  An env's ten agent
    images share their conv pre-activations, so one float32 decision of the device's shared trunk is up to ten candidates here.

    Returns (residual = got - ref - sum_{j in used} D_j as a dict of blocks, the 0/1 vector `used` over the candidates, the
    candidate list): a correct float32 gradient leaves a residual at float32 level with a handful of candidates used."""
    N = states.shape[0]
    amb = relu_ambiguous(p, states, eps)
    r = got_flat.astype(np.float64) - flatten_params(g_ref)
    if not amb:
        return unflatten_params(r), np.zeros(0), amb
    cols = []
    cache = {}
    for s, layer, idx, _ in amb:
        if s not in cache:      # one forward pass per sample; every flip of that sample is one more backward pass
            sl = slice(s, s + 1)
            mu, sigma, vs, c = conv_forward(p, states[sl], scale, keep=True)
            _, _, _, dmu, dsigma, dvs = gaussian_loss_terms(mu, sigma, actions[sl], advantages[sl], critic_target[sl], vs, beta, scale)
            m = {k: c[k] > 0 for k in RELU_LAYERS}
            cache = {s: (mu, sigma, c, dmu, dsigma, dvs, m, flatten_params(_conv_backward(p, c, mu, sigma, dmu, dsigma, dvs, m, scale)))}
        mu, sigma, c, dmu, dsigma, dvs, m, base = cache[s]
        m[layer].reshape(-1)[idx] ^= True
        cols.append((flatten_params(_conv_backward(p, c, mu, sigma, dmu, dsigma, dvs, m, scale)) - base) / N)
        m[layer].reshape(-1)[idx] ^= True
    # every block counts relative to its own largest entry (the blocks' scales differ by 10^4, and the comparison that follows is
    # per block)
    w = flatten_params({n: np.full(np.shape(g_ref[n]), 1.0 / max(np.abs(g_ref[n]).max(), 1e-300)) for n, _ in CONV_PARAM_SHAPES})
    A = np.stack(cols, axis=1) * w[:, None]
    rw = r * w
    norm2 = (A * A).sum(axis=0)
    used = np.zeros(A.shape[1])
    free = norm2 > 0                          # an element without upstream gradient cannot be seen either way
    for _ in range(max_flips):
        if not free.any():
            break
        gain = 2.0 * (A.T @ rw) - norm2       # ||rw||^2 - ||rw - D_j||^2
        gain[~free] = -np.inf
        j = int(np.argmax(gain))
        if not gain[j] > 0.5 * norm2[j]:      # <r, D_j> > 0.75 |D_j|^2: the column is in the residual about once
            break
        used[j] = 1.0
        free[j] = False
        rw = rw - A[:, j]
    return unflatten_params(rw / w), used, amb


def clip_by_global_norm(grads_flat, clip_norm):
    """tf.clip_by_global_norm (actor_learner.py:52-57): g * clip / max(norm, clip)."""
    norm = np.sqrt(np.sum(grads_flat ** 2))
    return grads_flat * (clip_norm / max(norm, clip_norm)), norm


def adam_step(params_flat, grads_flat, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (TF 1.4): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t * m / (sqrt(v) + eps).  t starts at 1."""
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * grads_flat
    v = b2 * v + (1 - b2) * grads_flat ** 2
    return params_flat - lr_t * m / (np.sqrt(v) + eps), m, v


# ------------------------------------------------------------------------------------------ Flat net
def flat_param_shapes(static_size=2, temporal_size=2, static_hidden=32, rnn_hidden=32, num_actions=1):
    """FlatPolicyVNetwork variables (policy_v_network.py:207-244, a3c/estimators.py:18-28)."""
    H, S = rnn_hidden, static_hidden
    return [
        ("gru_gates_w", (temporal_size + H, 2 * H)), ("gru_gates_b", (2 * H,)),     # r,u gates; bias init 1.0
        ("gru_cand_w", (temporal_size + H, H)), ("gru_cand_b", (H,)),
        ("temporal_w", (H, 2 * H)), ("temporal_b", (2 * H,)),
        ("static1_w", (static_size, 2 * H)), ("static1_b", (2 * H,)),
        ("static2_w", (2 * H, H)), ("static2_b", (H,)),
        ("mu1_w", (3 * H, 2 * S)), ("mu1_b", (2 * S,)),
        ("mu2_w", (2 * S, S)), ("mu2_b", (S,)),
        ("mu3_w", (S, num_actions)), ("mu3_b", (num_actions,)),
        ("sig1_w", (3 * H, 2 * S)), ("sig1_b", (2 * S,)),
        ("sig2_w", (2 * S, S)), ("sig2_b", (S,)),
        ("sig3_w", (S, num_actions)), ("sig3_b", (num_actions,)),                   # bias init -1.0
        ("v1_w", (3 * H, 2 * S)), ("v1_b", (2 * S,)),
        ("v2_w", (2 * S, 1)), ("v2_b", (1,)),
    ]


def flat_init(seed=3, **kw):
    rng = np.random.RandomState(seed)
    p = {}
    for n, s in flat_param_shapes(**kw):
        if n.endswith("_w"):
            p[n] = glorot_uniform(rng, s)
        elif n == "gru_gates_b":
            p[n] = np.ones(s)        # GRUCell gate bias initialiser = 1.0
        elif n == "sig3_b":
            p[n] = -np.ones(s)       # policy_v_network.py:225
        else:
            p[n] = np.zeros(s)
    return p


def gru_last_state(p, history):
    """tf.nn.dynamic_rnn(GRUCell) with sequence_length = #rows whose max|x| > 0
    (a3c/estimators.py:11-23): state stops updating once t >= length.  history (N,T,D)."""
    N, T, D = history.shape
    H = p["gru_cand_b"].shape[0]
    length = np.sign(np.max(np.abs(history), axis=2)).sum(axis=1).astype(int)
    h = np.zeros((N, H))
    for t in range(T):
        x = history[:, t]
        gates = _sigmoid(np.concatenate([x, h], 1) @ p["gru_gates_w"] + p["gru_gates_b"])
        r, u = gates[:, :H], gates[:, H:]
        c = np.tanh(np.concatenate([x, r * h], 1) @ p["gru_cand_w"] + p["gru_cand_b"])
        h_new = u * h + (1 - u) * c
        h = np.where((t < length)[:, None], h_new, h)
    return h


def flat_forward(p, states, history, scale, ub=5.0, lb=-5.0):
    """FlatPolicyVNetwork forward (policy_v_network.py:207-244)."""
    h = gru_last_state(p, history)
    dense_temporal = np.maximum(h @ p["temporal_w"] + p["temporal_b"], 0)
    s1 = np.maximum(states @ p["static1_w"] + p["static1_b"], 0)
    s2 = np.maximum(s1 @ p["static2_w"] + p["static2_b"], 0)
    x = np.concatenate([dense_temporal, s2], axis=1)
    m = np.maximum(x @ p["mu1_w"] + p["mu1_b"], 0)
    m = np.tanh(m @ p["mu2_w"] + p["mu2_b"])
    mu = ((ub - lb) / 2.0) * np.tanh(m @ p["mu3_w"] + p["mu3_b"]) + (lb + ub) / 2.0
    s = np.maximum(x @ p["sig1_w"] + p["sig1_b"], 0)
    s = np.tanh(s @ p["sig2_w"] + p["sig2_b"])
    sigma = _sigmoid(s @ p["sig3_w"] + p["sig3_b"]) + 1e-3
    v = np.tanh(x @ p["v1_w"] + p["v1_b"])
    vs = scale * (v @ p["v2_w"] + p["v2_b"])[:, 0]
    return mu, sigma, vs


def flat_loss(p, states, history, actions, advantages, critic_target, scale):
    mu, sigma, vs = flat_forward(p, states, history, scale)
    loss, pl, cl, _, _, _ = gaussian_loss_terms(mu, sigma, actions, advantages, critic_target, vs, 0.0, scale,
                                                entropy_in_loss=False)
    return loss, pl, cl


def flat_loss_and_grads(p, states, history, actions, advantages, critic_target, scale, ub=5.0, lb=-5.0):
    """Analytic gradients of FlatPolicyVNetwork's loss (policy_v_network.py:228-251) incl. back-propagation
    through the length-masked GRU.  Returns loss, policy_loss, critic_loss_mean, grads, (mu, sigma, vs)."""
    N, T, D = history.shape
    H = p["gru_cand_b"].shape[0]
    length = np.sign(np.max(np.abs(history), axis=2)).sum(axis=1).astype(int)
    hs, rs, us, cs = [np.zeros((N, H))], [], [], []
    h = hs[0]
    for t in range(T):
        x = history[:, t]
        gates = _sigmoid(np.concatenate([x, h], 1) @ p["gru_gates_w"] + p["gru_gates_b"])
        r, u = gates[:, :H], gates[:, H:]
        c = np.tanh(np.concatenate([x, r * h], 1) @ p["gru_cand_w"] + p["gru_cand_b"])
        h = np.where((t < length)[:, None], u * h + (1 - u) * c, h)
        rs.append(r); us.append(u); cs.append(c); hs.append(h)
    dt_ = np.maximum(h @ p["temporal_w"] + p["temporal_b"], 0)
    s1 = np.maximum(states @ p["static1_w"] + p["static1_b"], 0)
    s2 = np.maximum(s1 @ p["static2_w"] + p["static2_b"], 0)
    x96 = np.concatenate([dt_, s2], axis=1)
    m1 = np.maximum(x96 @ p["mu1_w"] + p["mu1_b"], 0); m2 = np.tanh(m1 @ p["mu2_w"] + p["mu2_b"])
    tm = np.tanh(m2 @ p["mu3_w"] + p["mu3_b"]); mu = ((ub - lb) / 2.0) * tm + (lb + ub) / 2.0
    g1 = np.maximum(x96 @ p["sig1_w"] + p["sig1_b"], 0); g2 = np.tanh(g1 @ p["sig2_w"] + p["sig2_b"])
    sg = _sigmoid(g2 @ p["sig3_w"] + p["sig3_b"]); sigma = sg + 1e-3
    v1 = np.tanh(x96 @ p["v1_w"] + p["v1_b"]); vs = scale * (v1 @ p["v2_w"] + p["v2_b"])[:, 0]
    loss, pl, cl, dmu, dsigma, dvs = gaussian_loss_terms(mu, sigma, actions, advantages, critic_target, vs, 0.0, scale,
                                                         entropy_in_loss=False)
    g = {}

    def dense_bwd(name, x, dz):
        g[name + "_w"], g[name + "_b"] = x.T @ dz, dz.sum(0)
        return dz @ p[name + "_w"].T
    dz = dmu * ((ub - lb) / 2.0) * (1 - tm ** 2)
    dm2 = dense_bwd("mu3", m2, dz) * (1 - m2 ** 2)
    dm1 = dense_bwd("mu2", m1, dm2) * (m1 > 0)
    dx = dense_bwd("mu1", x96, dm1)
    dz = dsigma * sg * (1 - sg)
    dg2 = dense_bwd("sig3", g2, dz) * (1 - g2 ** 2)
    dg1 = dense_bwd("sig2", g1, dg2) * (g1 > 0)
    dx = dx + dense_bwd("sig1", x96, dg1)
    dzv = (dvs * scale)[:, None]
    dv1 = dense_bwd("v2", v1, dzv) * (1 - v1 ** 2)
    dx = dx + dense_bwd("v1", x96, dv1)
    ddt, ds2 = dx[:, :2 * H] * (dt_ > 0), dx[:, 2 * H:] * (s2 > 0)
    ds1 = dense_bwd("static2", s1, ds2) * (s1 > 0)
    dense_bwd("static1", states, ds1)
    dh = dense_bwd("temporal", h, ddt)
    g["gru_gates_w"] = np.zeros_like(p["gru_gates_w"]); g["gru_gates_b"] = np.zeros_like(p["gru_gates_b"])
    g["gru_cand_w"] = np.zeros_like(p["gru_cand_w"]); g["gru_cand_b"] = np.zeros_like(p["gru_cand_b"])
    for t in reversed(range(T)):
        act = (t < length)[:, None]
        hp, r, u, c = hs[t], rs[t], us[t], cs[t]
        x = history[:, t]
        dhn = np.where(act, dh, 0.0)
        du, dc, dh_keep = dhn * (hp - c), dhn * (1 - u), dhn * u
        dzc = dc * (1 - c ** 2)
        xin = np.concatenate([x, r * hp], 1)
        g["gru_cand_w"] += xin.T @ dzc; g["gru_cand_b"] += dzc.sum(0)
        drh = (dzc @ p["gru_cand_w"].T)[:, D:]
        dr = drh * hp
        dzg = np.concatenate([dr * r * (1 - r), du * u * (1 - u)], 1)
        xin = np.concatenate([x, hp], 1)
        g["gru_gates_w"] += xin.T @ dzg; g["gru_gates_b"] += dzg.sum(0)
        dh_prev = dh_keep + drh * r + (dzg @ p["gru_gates_w"].T)[:, D:]
        dh = np.where(act, dh_prev, dh)
    return loss, pl, cl, g, (mu, sigma, vs)


def numeric_grad(f, p, names, eps=1e-6, max_per=6, seed=0):
    """Central finite differences on a few random entries of each named parameter."""
    rng = np.random.RandomState(seed)
    out = {}
    for n in names:
        idxs = [tuple(rng.randint(0, s) for s in p[n].shape) for _ in range(max_per)]
        vals = []
        for idx in idxs:
            old = p[n][idx]
            p[n][idx] = old + eps; fp = f(p)
            p[n][idx] = old - eps; fm = f(p)
            p[n][idx] = old
            vals.append((idx, (fp - fm) / (2 * eps)))
        out[n] = vals
    return out


# ------------------------------------------------------------------------------------------ Field net
# ConvPolicyVFieldNetwork (policy_v_network.py:83-191; placeholders networks.py:170-190).  PARITY UNPINNED like the two nets
# above (the reference's only test of it is a shape test at 32x32x3, 5 filters, 2 conv layers, 3 actions:
# tests/estimators_tests.py:152-215).  use_rnn is False in the reference (:88): the history input is never consumed.
def field_param_shapes(height=32, width=32, channels=3, filters=5, conv_layers=2, num_actions=3, fc_hidden=32):
    """tf.trainable_variables() creation order: the conv layers and dense1/dense2 of 'process_input', the four 'policy'
    layers, the three 'v_s' layers."""
    fh, fw = int(height / (2 ** conv_layers)), int(width / (2 ** conv_layers))
    shapes, cin = [], channels
    for i in range(conv_layers):
        shapes += [("conv%d_w" % i, (3, 3, cin, filters)), ("conv%d_b" % i, (filters,))]
        cin = filters
    hwa = height * width * num_actions
    shapes += [("dense1_w", (fh * fw * filters, 2 * fc_hidden)), ("dense1_b", (2 * fc_hidden,)),
               ("dense2_w", (2 * fc_hidden, fc_hidden)), ("dense2_b", (fc_hidden,)),
               ("pol1_w", (fc_hidden, 2 * fc_hidden)), ("pol1_b", (2 * fc_hidden,)),
               ("pol2_w", (2 * fc_hidden, 2 * hwa)), ("pol2_b", (2 * hwa,)),
               ("mu_w", (2 * hwa, hwa)), ("mu_b", (hwa,)), ("sigma_w", (2 * hwa, hwa)), ("sigma_b", (hwa,)),
               ("v1_w", (fc_hidden, 2 * fc_hidden)), ("v1_b", (2 * fc_hidden,)), ("v2_w", (2 * fc_hidden, fc_hidden)), ("v2_b", (fc_hidden,)),
               ("v3_w", (fc_hidden, 1)), ("v3_b", (1,))]
    return shapes


def _conv_same3(x, w, b):
    """tf.layers.Conv2D(kernel_size=3, padding='same'): zero padding of one pixel on every side."""
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    return _conv(xp, w, b, 1)


def _maxpool2(a):
    """tf.layers.MaxPooling2D(2, 2) on even H, W.  Returns pooled (N,H/2,W/2,C) and the argmax (0..3 = dy*2+dx; the FIRST maximum,
    as the gradient of the TF op routes ties)."""
    N, H, W, C = a.shape
    win = a.reshape(N, H // 2, 2, W // 2, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(N, H // 2, W // 2, 4, C)
    idx = win.argmax(axis=3)
    return np.take_along_axis(win, idx[:, :, :, None, :], axis=3)[:, :, :, 0, :], idx


def field_forward(p, states, positions, scale, conv_layers=2, keep=False):
    """states (N,H,W,C), positions (N,2) int [(height_idx, width_idx)] -> mu (N,A), sigma (N,A), vs (N,)."""
    N, H, W, _ = states.shape
    c = {"x": [], "cols": [], "z": [], "idx": []}
    x = states
    for i in range(conv_layers):
        c["x"].append(x)
        z, cols = _conv_same3(x, p["conv%d_w" % i], p["conv%d_b" % i])
        pooled, idx = _maxpool2(np.maximum(z, 0))
        c["cols"].append(cols); c["z"].append(z); c["idx"].append(idx)
        x = pooled
    flat = x.reshape(N, -1)                                   # tf.reshape of NHWC: (h, w, f)
    d1 = np.maximum(flat @ p["dense1_w"] + p["dense1_b"], 0)
    d2 = np.maximum(d1 @ p["dense2_w"] + p["dense2_b"], 0)   # processed_state
    p1 = np.maximum(d2 @ p["pol1_w"] + p["pol1_b"], 0)
    p2 = np.maximum(p1 @ p["pol2_w"] + p["pol2_b"], 0)
    A = p["mu_w"].shape[1] // (H * W)
    # mus/sigmas are Dense(H*W*A) reshaped (N,H,W,A) and gathered at the agent's position (:140-152): only A columns per sample
    col = ((positions[:, 0] * W + positions[:, 1]) * A)[:, None] + np.arange(A)[None]      # (N, A)
    zmu = np.einsum("nk,kna->na", p2, p["mu_w"][:, col]) + p["mu_b"][col]
    zsg = np.einsum("nk,kna->na", p2, p["sigma_w"][:, col]) + p["sigma_b"][col]
    mu, sigma = np.tanh(zmu), _sigmoid(zsg)
    v1 = np.maximum(d2 @ p["v1_w"] + p["v1_b"], 0)
    v2 = np.maximum(v1 @ p["v2_w"] + p["v2_b"], 0)
    zv = (v2 @ p["v3_w"] + p["v3_b"])[:, 0]
    vs = -scale * softplus(zv)
    if keep:
        c.update(final=x, flat=flat, d1=d1, d2=d2, p1=p1, p2=p2, col=col, mu=mu, sigma=sigma, v1=v1, v2=v2, zv=zv)
        return mu, sigma, vs, c
    return mu, sigma, vs


def field_loss_and_grads(p, states, positions, actions, advantages, critic_target, beta, scale, conv_layers=2):
    """loss (policy_v_network.py:154-173: the N1 loss, entropy term included) and d loss / d params, float64."""
    mu, sigma, vs, c = field_forward(p, states, positions, scale, conv_layers, keep=True)
    loss, pl, cl, dmu, dsigma, dvs = gaussian_loss_terms(mu, sigma, actions, advantages, critic_target, vs, beta, scale)
    N = states.shape[0]
    g = {k: np.zeros_like(v) for k, v in p.items()}
    dzmu, dzsg = dmu * (1 - mu ** 2), dsigma * sigma * (1 - sigma)
    dp2 = np.zeros_like(c["p2"])
    for n in range(N):            # samples in order: two agents on one pixel add into the same columns
        cols = c["col"][n]
        g["mu_w"][:, cols] += np.outer(c["p2"][n], dzmu[n]); g["mu_b"][cols] += dzmu[n]
        g["sigma_w"][:, cols] += np.outer(c["p2"][n], dzsg[n]); g["sigma_b"][cols] += dzsg[n]
        dp2[n] = p["mu_w"][:, cols] @ dzmu[n] + p["sigma_w"][:, cols] @ dzsg[n]
    dp2 *= c["p2"] > 0
    g["pol2_w"], g["pol2_b"] = c["p1"].T @ dp2, dp2.sum(0)
    dp1 = (dp2 @ p["pol2_w"].T) * (c["p1"] > 0)
    g["pol1_w"], g["pol1_b"] = c["d2"].T @ dp1, dp1.sum(0)
    dzv = (dvs * (-scale) * _sigmoid(c["zv"]))[:, None]
    g["v3_w"], g["v3_b"] = c["v2"].T @ dzv, dzv.sum(0)
    dv2 = (dzv @ p["v3_w"].T) * (c["v2"] > 0)
    g["v2_w"], g["v2_b"] = c["v1"].T @ dv2, dv2.sum(0)
    dv1 = (dv2 @ p["v2_w"].T) * (c["v1"] > 0)
    g["v1_w"], g["v1_b"] = c["d2"].T @ dv1, dv1.sum(0)
    dd2 = (dp1 @ p["pol1_w"].T + dv1 @ p["v1_w"].T) * (c["d2"] > 0)
    g["dense2_w"], g["dense2_b"] = c["d1"].T @ dd2, dd2.sum(0)
    dd1 = (dd2 @ p["dense2_w"].T) * (c["d1"] > 0)
    g["dense1_w"], g["dense1_b"] = c["flat"].T @ dd1, dd1.sum(0)
    dx = (dd1 @ p["dense1_w"].T).reshape(c["final"].shape)
    for i in reversed(range(conv_layers)):
        z, idx = c["z"][i], c["idx"][i]
        Nn, Hh, Ww, F = z.shape
        # max-pool gradient to the argmax of each window, then ReLU
        dwin = np.zeros((Nn, Hh // 2, Ww // 2, 4, F))
        np.put_along_axis(dwin, idx[:, :, :, None, :], dx[:, :, :, None, :], axis=3)
        da = dwin.reshape(Nn, Hh // 2, Ww // 2, 2, 2, F).transpose(0, 1, 3, 2, 4, 5).reshape(Nn, Hh, Ww, F)
        dz = da * (z > 0)
        w = p["conv%d_w" % i]
        g["conv%d_w" % i] = (c["cols"][i].reshape(-1, c["cols"][i].shape[-1]).T @ dz.reshape(-1, F)).reshape(w.shape)
        g["conv%d_b" % i] = dz.sum(axis=(0, 1, 2))
        xs = c["x"][i].shape
        dxp = _col2im(dz @ w.reshape(-1, F).T, (xs[0], xs[1] + 2, xs[2] + 2, xs[3]), 3, 3, 1)
        dx = dxp[:, 1:-1, 1:-1, :]
    return loss, pl, cl, g, (mu, sigma, vs)
