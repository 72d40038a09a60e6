"""GPU parity of ConvSingleAgentPolicyNetwork against the float64 restatement oracle/nets.py ABOVE ONE ROW TILE (round 4).

The support / tap / row-list machinery of net_patch.inc and net_shared.inc (729-key patch sort with a 25-bit union per 128 sorted
rows, slot tap-class sort per 256 rows, trunk row lists, chunk union mask, closed-form background terms) only does anything
interesting when a chunk holds several row tiles; tests/test_gpu_net.py compares with the oracle at <= 80 samples and checks
the larger sizes HIP-against-HIP.  Here 640 and 1 280 samples (64 / 128 envs; chunks of 500 samples, so the last chunk is
ragged and two to three stream lanes are busy) go against the oracle directly, in both support regimes:

  rim       the engine's own state after a few random steps: half of the agents sit in the outer bins of the observation grid
            (1x1 .. 3x3 slot rectangles, 22-26 % patch support: the bench workload's geometry)
  interior  every agent at positions 20..60 (3x3 slots, 5x5 supports, 9 live taps) over the same locust bins

Reference: fed_gym/agents/paac/policy_v_network.py:14-66 (forward, loss), paac.py:302-387 (the rollout the gradient step consumes).
Tolerances: forward 2e-5 relative (values: 2e-5 of the scale-1000 range), every gradient block <= 2e-5 of its largest entry once the
ReLU inputs within float32 round-off of zero are accounted for (_check_grads).
Network numerics are 'parity unpinned' wrt TensorFlow (absent); see DESIGN.md section 4."""
import numpy as np
import pytest

from oracle import nets as NN
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _states(lb, ab, pos):
    out = []
    for e in range(lb.shape[0]):
        l, a = lb[e].astype(int), ab[e].astype(int)
        l[l[:, 0] == 255] = -1; a[a[:, 0] == 255] = -1
        out.append(O.swarm_local_states(O.swarm_grid_from_compact(l, a, 84), pos[e]))
    return np.concatenate(out).astype(np.float32).astype(np.float64)      # the TF placeholder is float32


def _biased_params(seed):
    from goldsrl import _ffi_net
    rng = np.random.RandomState(seed)
    p = NN.unflatten_params(_ffi_net.glorot_uniform_flat(seed=seed).astype(np.float64))
    for k in p:
        if k.endswith("_b"):
            p[k] = rng.normal(size=p[k].shape) * 0.05      # the background terms of the trunk vanish with b1 = 0
    flat = NN.flatten_params(p).astype(np.float32)
    return flat, NN.unflatten_params(flat.astype(np.float64))


def _observations(eng, E, regime, rng):
    for _ in range(3):
        eng.step(O.swarm_transform_actions(rng.normal(size=(E, 10, 2)).astype(np.float32)))
    lb, ab, pos = eng.read("locust_bins"), eng.read("agent_bins"), eng.read("positions")
    if regime == "interior":
        pos = rng.randint(20, 61, size=(E, 10, 2)).astype(np.uint8)
        ab = (pos - 1).astype(np.uint8)      # the agent-density bin behind a one-hot at pos (quirk Q2: the one-hot is offset +1, +1)
    return lb, ab, pos


RAW_BOUND = 5e-3      # of a block's largest entry: what ONE flipped ReLU derivative of one sample can move at >= 600 samples (measured
                      # 4e-4 .. 1.8e-3 on conv1_w); anything above is an error whatever the explainer says
MAX_FLIPS = 32        # candidates that may be used: one float32 decision of the device's shared trunk is up to ten agent-samples here


def _check_grads(got_flat, ref, p, states, act, adv, y, tol=2e-5, tag=""):
    """Every gradient block within tol of its largest entry -- ten times tighter than tests/test_gpu_net.py's bar -- AFTER the ReLU
    decisions float32 cannot be held to are accounted for: among ~10^7 pre-activations of a 1 000-sample batch a handful sit within
    float32 round-off of zero, the device may put them on the other side, and each such element moves whole gradient entries (the
    one-hot rows of conv1_w by 1e-3 of the block's maximum).  oracle.nets.explain_by_relu_flips subtracts the EXACT effect of a small
    set of such candidates (|z| < 2e-6 of the layer's range), each taken whole or not at all -- nothing is fitted (round 5; VERDICT
    r4 #6 / advisor r4: the least-squares fit over hundreds of free coefficients could have absorbed a genuine error); what is left
    must be float32 rounding.  The raw difference is bounded on its own."""
    got = NN.unflatten_params(got_flat.astype(np.float64))
    raw = {}
    for name, _ in NN.CONV_PARAM_SHAPES:
        scale = np.abs(ref[name]).max()
        assert scale > 0, (tag, name)
        raw[name] = np.abs(got[name] - ref[name]).max() / scale
    assert max(raw.values()) < RAW_BOUND, (tag, raw)
    if max(raw.values()) < tol:
        print("%s: raw worst %.1e (%s): no ReLU decision differs" % (tag, max(raw.values()), max(raw, key=raw.get)))
        return
    res, used, amb = NN.explain_by_relu_flips(p, states, act.astype(np.float64), adv.astype(np.float64), y.astype(np.float64), 0.02, 1000.0,
                                              got_flat, ref, max_flips=MAX_FLIPS)
    left = {k: np.abs(res[k]).max() / np.abs(ref[k]).max() for k in res}
    taken = [(amb[i][0], amb[i][1], amb[i][2]) for i in np.flatnonzero(used)]
    print("%s: raw worst %.1e (%s); %d ambiguous ReLU inputs, %d taken whole (coefficient 1) %s, residual worst %.1e" %
          (tag, max(raw.values()), max(raw, key=raw.get), len(amb), len(taken), taken[:12], max(left.values())))
    assert 0 < len(amb) < 600, (tag, len(amb))
    assert set(np.unique(used)) <= {0.0, 1.0} and 0 < len(taken) <= MAX_FLIPS, (tag, taken)
    # the candidates taken are few PHYSICAL decisions: elements of the env-level trunk come as (up to) the env's ten agent-samples
    physical = {(s // 10 if layer in ("a1", "a2", "a3") else s, layer, idx) for s, layer, idx in taken}
    assert len(physical) <= 8, (tag, sorted(physical))
    bad = {k: v for k, v in left.items() if not v < tol}
    assert not bad, (tag, bad, raw)


@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("regime", ["rim", "interior"])
def test_forward_and_gradients_match_the_oracle_above_one_row_tile(E, regime):
    from goldsrl import _ffi, _ffi_net
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=21)
    eng.reset()
    rng = np.random.RandomState(100 + E)
    lb, ab, pos = _observations(eng, E, regime, rng)
    if regime == "rim":      # the premise of the case: the engine's agents do sit on the rim (and inside for the other regime)
        assert ((pos.astype(int) < 4) | (pos.astype(int) > 79)).any(axis=2).mean() > 0.1
    else:
        assert pos.min() >= 20 and pos.max() <= 60
    flat, p = _biased_params(7)
    states = _states(lb, ab, pos)
    n = E * 10
    act = (rng.normal(size=(n, 2)) * 0.7).astype(np.float32)
    adv = (rng.normal(size=n) * 0.02).astype(np.float32)
    y = (-rng.rand(n) * 400).astype(np.float32)
    mu, sigma, vs = NN.conv_forward(p, states, 1000.0)
    loss, pl, cl, g, _ = NN.conv_loss_and_grads(p, states, act.astype(np.float64), adv.astype(np.float64), y.astype(np.float64), 0.02, 1000.0)
    net = _ffi_net.ConvNet(eng, max_chunk_samples=500)      # 50 envs per chunk: 500 + 140 or 500 + 500 + 280 samples
    net.set_params(flat)
    out = net.predict_obs(lb, ab, pos)
    np.testing.assert_allclose(out["mu"], mu, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["sigma"], sigma, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["vs"], vs, rtol=2e-5, atol=2e-5 * 1000.0)
    st = net.train_obs(lb, ab, pos, act, adv, y, lr=0.0, apply_update=False)
    np.testing.assert_allclose(st["loss"], loss, rtol=1e-4)
    np.testing.assert_allclose(st["policy_loss"], pl, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(st["critic_loss_mean"], cl, rtol=1e-4)
    g1 = net.get_grads().copy()
    _check_grads(g1, g, p, states, act, adv, y, tag=(E, regime))
    gn = np.sqrt(sum((g[k] ** 2).sum() for k in g))
    np.testing.assert_allclose(st["global_norm"], gn, rtol=1e-4)
    net.train_obs(lb, ab, pos, act, adv, y, lr=0.0, apply_update=False)
    assert np.array_equal(g1, net.get_grads())      # bitwise reproducible at this size too
    net.close()
    eng.close()


@pytest.mark.parametrize("regime", ["rim", "interior"])
def test_rollout_gradient_matches_the_oracle_at_64_envs(regime):
    """grl_net_rollout(T = 3) + grl_net_train_rollout_grads at 64 envs (1 920 samples, chunks of 500: four per step, the
    rollout-resident activations in use) against the oracle's pieces: the stored observations, raw actions, returns and
    advantages of the rollout fed to the float64 loss / gradient of the same parameters (paac.py:360-387: time-major
    flatten, adv / scale, mean over T * B).  interior (round 5): the agents start inside the observation box, near the swarm -- where
    a policy takes them within 100 updates (profiles/r04_training_geometry.json) -- so every step of the rollout runs 3 x 3 slot
    rectangles, 5 x 5 supports and a full union mask through the resident-activation path."""
    from goldsrl import _ffi, _ffi_net
    E, T = 64, 3
    B = E * 10
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=33)
    eng.reset()
    if regime == "interior":
        rng = np.random.RandomState(5)
        x = eng.get_state("SWARM_X")
        xa = np.empty((E, 10, 2))
        xa[:, :, 0] = x[:, :, 0].mean(axis=1)[:, None] + rng.uniform(-1.2, 0.0, size=(E, 10))
        xa[:, :, 1] = rng.uniform(1.5, 4.3, size=(E, 10))
        eng.set_state("SWARM_XA", xa)
        eng.observe()
        eng.wait()
    flat, p = _biased_params(9)
    net = _ffi_net.ConvNet(eng, max_chunk_samples=500)
    net.set_params(flat)
    net.rollout(T, 0)
    eng.wait()
    lb = net.read_rollout("locust_bins", (T, E, 80, 2), np.uint8)
    ab = net.read_rollout("agent_bins", (T, E, 10, 2), np.uint8)
    ps = net.read_rollout("positions", (T, E, 10, 2), np.uint8)
    inside = ((ps.astype(int) >= 8) & (ps.astype(int) <= 75)).all(axis=3).mean()
    assert inside > 0.95 if regime == "interior" else inside < 0.9, (regime, inside)      # the premise of the case
    acts = net.read_rollout("actions", (T, B, 2)); vals = net.read_rollout("values", (T, B)); rews = net.read_rollout("rewards", (T, B))
    yy = net.read_rollout("y", (T, B)); adv = net.read_rollout("adv", (T, B)); boot = net.read_rollout("boot", (B,))
    states = np.concatenate([_states(lb[t], ab[t], ps[t]) for t in range(T)])
    mu, sigma, vs = NN.conv_forward(p, states, 1000.0)
    np.testing.assert_allclose(vals.reshape(-1), vs, rtol=2e-5, atol=2e-5 * 1000.0)
    oy, oadv = O.nstep_returns(rews.astype(np.float64), vals, boot, 0.99)
    np.testing.assert_allclose(yy, oy, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(adv, oadv / 1000.0, rtol=1e-5, atol=1e-6)
    loss, pl, cl, g, _ = NN.conv_loss_and_grads(p, states, acts.reshape(-1, 2).astype(np.float64), adv.reshape(-1).astype(np.float64),
                                                yy.reshape(-1).astype(np.float64), 0.02, 1000.0)
    st = net.train_rollout_grads()
    np.testing.assert_allclose(st["loss"], loss, rtol=1e-4)
    _check_grads(net.get_grads(), g, p, states, acts.reshape(-1, 2), adv.reshape(-1), yy.reshape(-1), tag="rollout-" + regime)
    net.close()
    eng.close()


def test_workspace_left_by_other_geometry_is_never_read():
    """dza cells outside an agent's slot rectangle are not written and a2sh / a3sh are filled only under the chunk's union mask
    (net_shared.inc); the gathers' predicates are what keeps data of EARLIER chunks out of the GEMMs.  A net whose workspace was
    used on interior agents (3x3 slot rectangles, full union mask, large values) must give, on rim agents with 1x1 .. 3x3
    rectangles and out-of-box agents, bit for bit the gradient a fresh net gives -- and the oracle's."""
    from goldsrl import _ffi, _ffi_net
    E = 60
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=5)
    eng.reset()
    rng = np.random.RandomState(77)
    lbA, abA, posA = _observations(eng, E, "interior", rng)
    lbA = rng.randint(0, 84, size=(E, 80, 2)).astype(np.uint8)      # locusts everywhere: every trunk row affected
    edge = [0, 1, 2, 3, 80, 81, 82, 83]
    lbB = rng.randint(0, 84, size=(E, 80, 2)).astype(np.uint8)
    lbB[:, :, 1] %= 20                                               # a strip: the union mask is a proper subset
    posB = np.zeros((E, 10, 2), np.uint8)
    for e in range(E):
        for a in range(10):
            k = (a + e) % 3
            posB[e, a] = (rng.choice(edge), rng.choice(edge)) if k == 0 else (rng.choice(edge), rng.randint(0, 20)) if k == 1 else \
                (rng.randint(0, 84), rng.randint(0, 20))
    abB = posB.copy()
    abB[::7, :2] = 255                                               # agents outside the box: no density entry, one-hot kept (quirk Q3)
    flat, p = _biased_params(11)
    n = E * 10
    actA = (rng.normal(size=(n, 2)) * 0.7).astype(np.float32); advA = (rng.normal(size=n) * 30).astype(np.float32)      # large gradients left behind
    yA = (-rng.rand(n) * 400).astype(np.float32)
    actB = (rng.normal(size=(n, 2)) * 0.7).astype(np.float32); advB = (rng.normal(size=n) * 0.02).astype(np.float32)
    yB = (-rng.rand(n) * 400).astype(np.float32)
    fresh = _ffi_net.ConvNet(eng, max_chunk_samples=250)      # 25 envs per chunk: three chunks, the last ragged
    fresh.set_params(flat)
    outF = fresh.predict_obs(lbB, abB, posB)
    fresh.train_obs(lbB, abB, posB, actB, advB, yB, lr=0.0, apply_update=False)
    gF = fresh.get_grads().copy()
    fresh.close()
    dirty = _ffi_net.ConvNet(eng, max_chunk_samples=250)
    dirty.set_params(flat)
    dirty.predict_obs(lbA, abA, posA)
    dirty.train_obs(lbA, abA, posA, actA, advA, yA, lr=0.0, apply_update=False)
    outD = dirty.predict_obs(lbB, abB, posB)
    dirty.train_obs(lbB, abB, posB, actB, advB, yB, lr=0.0, apply_update=False)
    gD = dirty.get_grads().copy()
    dirty.close()
    for k in ("mu", "sigma", "vs"):
        assert np.array_equal(outF[k], outD[k]), k
    assert np.isfinite(gF).all() and np.array_equal(gF, gD)
    states = _states(lbB, abB, posB)
    _, _, _, g, _ = NN.conv_loss_and_grads(p, states, actB.astype(np.float64), advB.astype(np.float64), yB.astype(np.float64), 0.02, 1000.0)
    _check_grads(gD, g, p, states, actB, advB, yB, tag="dirty")
    eng.close()
