"""GPU parity of FlatPolicyVNetwork (GRU + MLP, HIP, one lane per sample) against oracle/nets.py (float64,
'parity unpinned' wrt TensorFlow) with shared weights; and the flat PAAC rollout on the Solow engine."""
import numpy as np
import pytest

from oracle import nets as NN
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _net(kind_kw, n_env=8, **cfg):
    from goldsrl import _ffi, _ffi_flat
    eng = _ffi.Engine(_ffi.ENV_SOLOW, n_env, seed=5, **kind_kw)
    eng.reset()
    net = _ffi_flat.FlatNet(eng, **cfg)
    return eng, net


def _params(net, static_size=2, temporal_size=2, num_actions=1, seed=4):
    from goldsrl import _ffi_flat
    shapes = _ffi_flat.flat_param_shapes(static_size, temporal_size, num_actions)
    assert [tuple(s) for _, s in shapes] == [tuple(s) for _, s in NN.flat_param_shapes(static_size, temporal_size, 32, 32, num_actions)]
    flat = _ffi_flat.default_init_flat(seed, static_size=static_size, temporal_size=temporal_size, num_actions=num_actions)
    rng = np.random.RandomState(seed)
    flat = flat + (rng.normal(size=flat.size) * 0.05).astype(np.float32)      # move biases off their init
    net.set_params(flat)
    assert net.num_params == flat.size
    return NN.unflatten_params(flat.astype(np.float64), NN.flat_param_shapes(static_size, temporal_size, 32, 32, num_actions))


def _samples(n, S0, D, T, A, seed=0):
    rng = np.random.RandomState(seed)
    states = (rng.normal(size=(n, S0)) * 0.3).astype(np.float32)
    hist = (rng.normal(size=(n, T, D)) * 0.3).astype(np.float32)
    for i in range(n):
        hist[i, 1 + i % T:] = 0          # ragged lengths 1..T (zero rows end the sequence)
    act = rng.normal(size=(n, A)).astype(np.float32) * 2
    adv = (rng.normal(size=n) * 0.3).astype(np.float32)
    y = (rng.normal(size=n) * 40).astype(np.float32)
    return states, hist, act, adv, y


@pytest.mark.parametrize("S0,D,T,A", [(2, 2, 5, 1), (33, 33, 20, 16)])     # Solow defaults; TradeAR1-16 shapes (train_trade.py:38-40,119)
def test_flat_forward_and_gradients_match_oracle(S0, D, T, A):
    n = 150       # three wave-groups, last one partial
    eng, net = _net({}, static_size=S0, temporal_size=D, rnn_length=T, num_actions=A, max_samples=256, scale=100.0)
    p = _params(net, S0, D, A)
    states, hist, act, adv, y = _samples(n, S0, D, T, A)
    out = net.predict(states, hist)
    mu, sigma, vs = NN.flat_forward(p, states.astype(np.float64), hist.astype(np.float64), 100.0)
    np.testing.assert_allclose(out["mu"], mu, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out["sigma"], sigma, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["vs"], vs, rtol=2e-5, atol=2e-4)
    assert (np.abs(out["mu"]) <= 5).all() and (out["sigma"] > 1e-3).all()
    st = net.train(states, hist, act, adv, y, lr=0.0, apply_update=False)
    loss, pl, cl, g, _ = NN.flat_loss_and_grads(p, states.astype(np.float64), hist.astype(np.float64), act.astype(np.float64),
                                                adv.astype(np.float64), y.astype(np.float64), 100.0)
    np.testing.assert_allclose([st["loss"], st["policy_loss"], st["critic_loss_mean"]], [loss, pl, cl], rtol=1e-4, atol=1e-6)
    shapes = NN.flat_param_shapes(S0, D, 32, 32, A)
    got = NN.unflatten_params(net.get_grads().astype(np.float64), shapes)
    for name, _ in shapes:
        err = np.abs(got[name] - g[name]).max() / (np.abs(g[name]).max() + 1e-12)
        assert err < 2e-4, (name, err)
    np.testing.assert_allclose(st["global_norm"], np.sqrt(sum((v ** 2).sum() for v in g.values())), rtol=1e-4)
    first = net.get_grads()
    net.train(states, hist, act, adv, y, lr=0.0, apply_update=False)
    assert np.array_equal(first, net.get_grads())          # fixed-order reduction: reproducible


def test_flat_adam_steps_match_oracle():
    eng, net = _net({}, max_samples=256, clip_norm=0.05)
    p = _params(net)
    shapes = NN.flat_param_shapes()
    states, hist, act, adv, y = _samples(100, 2, 2, 5, 1, seed=3)
    pf = NN.flatten_params(p, shapes); flat0 = pf.copy(); m = np.zeros_like(pf); v = np.zeros_like(pf)
    for step in range(1, 4):
        _, _, _, g, _ = NN.flat_loss_and_grads(NN.unflatten_params(pf, shapes), states.astype(np.float64), hist.astype(np.float64),
                                               act.astype(np.float64), adv.astype(np.float64), y.astype(np.float64), 100.0)
        gf, norm = NN.clip_by_global_norm(NN.flatten_params(g, shapes), 0.05)
        assert norm > 0.05
        pf, m, v = NN.adam_step(pf, gf, m, v, step, 1e-3)
        st = net.train(states, hist, act, adv, y, lr=1e-3)
        np.testing.assert_allclose(st["global_norm"], norm, rtol=2e-4)
        np.testing.assert_allclose(net.get_params().astype(np.float64) - flat0, pf - flat0, rtol=0, atol=3e-5 * step)


@pytest.mark.parametrize("E", [256, 4096])      # 4 096 envs x 20 steps = BASELINE config 2
def test_flat_paac_rollout_on_solow_engine(E):
    T = 20
    eng, net = _net({"max_episode_steps": 8}, n_env=E, max_samples=E * T)
    p = _params(net)
    obs0, hist0 = eng.read("obs"), eng.read("history")
    pred = net.predict_env()
    mu, sigma, vs = NN.flat_forward(p, obs0.astype(np.float64), hist0.astype(np.float64), 100.0)
    np.testing.assert_allclose(pred["vs"], vs, rtol=2e-5, atol=2e-4)
    net.rollout(T)
    eng.wait()
    acts, vals, rews, masks = (net.read_rollout(k, (T, E)) for k in ("actions", "values", "rewards", "masks"))
    yy, adv, boot = net.read_rollout("y", (T, E)), net.read_rollout("adv", (T, E)), net.read_rollout("boot", (E,))
    assert np.array_equal(vals[0], pred["vs"])
    eps = O.normal_pair(O.rng_block(5, np.arange(E), 0, 17, 0))[0]
    np.testing.assert_allclose(acts[0], pred["mu"][:, 0].astype(np.float64) + pred["sigma"][:, 0].astype(np.float64) * eps, rtol=1e-6, atol=1e-6)
    # TimeLimit(8): every env finishes at steps 7 and 15 -> masks are 0 there and the return restarts
    assert (masks[7] == 0).all() and (masks[15] == 0).all() and masks.sum() == (T - 2) * E
    oy, oadv = O.nstep_returns(O.rescale_reward(rews).astype(np.float64), vals, boot, 0.99, masks.astype(np.float64))
    np.testing.assert_allclose(yy, oy, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv, oadv / 100.0, rtol=1e-5, atol=1e-6)
    # stored (state, history) pairs reproduce the stored values through predict()
    st_ = net.read_rollout("states", (T, E, 2)); hi_ = net.read_rollout("histories", (T, E, 5, 2))
    np.testing.assert_array_equal(net.predict(st_[9], hi_[9])["vs"], vals[9])
    stats = net.train_rollout(1e-4)
    assert all(np.isfinite(list(stats.values()))) and stats["global_norm"] > 0
    # one more rollout with the updated weights still runs (values change)
    net.rollout(T); eng.wait()
    assert not np.array_equal(net.read_rollout("values", (T, E)), vals)


def test_trade_paac_rollout_gru_policy():
    """BASELINE config 5 shape: TradeAR1 n=16, GRU policy (static = temporal = 33, 16 actions, window 20)."""
    from goldsrl import _ffi, _ffi_flat
    E, T, n, R = 192, 6, 16, 20
    S = 1 + 2 * n
    eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=9, n_assets=n, rnn_length=R, max_episode_steps=4)
    eng.reset()
    net = _ffi_flat.FlatNet(eng, static_size=S, temporal_size=S, rnn_length=R, num_actions=n, max_samples=E * T, scale=100.0)
    p = _params(net, S, S, n)
    obs0 = eng.read("obs")
    pred = net.predict_env()
    hist0 = np.zeros((E, R, S)); hist0[:, 0] = obs0                 # one row after an explicit reset
    mu, sigma, vs = NN.flat_forward(p, obs0.astype(np.float64), hist0, 100.0)
    np.testing.assert_allclose(pred["mu"], mu, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(pred["vs"], vs, rtol=2e-5, atol=2e-4)
    net.rollout(T); eng.wait()
    acts = net.read_rollout("actions", (T, E, n)); vals = net.read_rollout("values", (T, E)); rews = net.read_rollout("rewards", (T, E))
    masks = net.read_rollout("masks", (T, E)); boot = net.read_rollout("boot", (E,))
    nh = np.empty((T, E), np.int32); net._check(net.lib.grl_fnet_read_rollout(net.n, b"nhist", _ffi._ptr(nh), nh.nbytes))
    st = net.read_rollout("states", (T, E, S))
    assert np.array_equal(vals[0], pred["vs"])
    # window rows: 0 (shown as 1) after the explicit reset, then 1,2,3, reset at the TimeLimit(4) -> 1, ...
    assert nh[:, 0].tolist() == [0, 1, 2, 3, 1, 2]
    assert (masks[3] == 0).all() and masks.sum() == (T - 1) * E
    # stored (state, #rows) pairs reproduce the stored values through the dense-history predict()
    t = 2
    hist = np.zeros((E, R, S), np.float32); hist[:, :2] = st[t][:, None]
    np.testing.assert_allclose(net.predict(st[t], hist)["vs"], vals[t], rtol=1e-6, atol=1e-5)
    # tanh-transformed actions were legal (no GRL_E_ACTION_RANGE), returns follow the masked/clipped PAAC rule
    y = net.read_rollout("y", (T, E)); adv = net.read_rollout("adv", (T, E))
    oy, oadv = O.nstep_returns(O.rescale_reward(rews).astype(np.float64), vals, boot, 0.99, masks.astype(np.float64))
    np.testing.assert_allclose(y, oy, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv, oadv / 100.0, rtol=1e-5, atol=1e-6)
    # gradients on the rollout's samples equal the oracle's on the equivalent dense histories
    stats = net.train_rollout(0.0)
    hist_all = np.zeros((T * E, R, S))
    flat_states = st.reshape(T * E, S).astype(np.float64)
    rows = np.minimum(np.maximum(nh.reshape(-1), 1), R)
    for r in range(R):
        hist_all[:, r] = np.where((r < rows)[:, None], flat_states, 0.0)
    loss, pl, cl, g, _ = NN.flat_loss_and_grads(p, flat_states, hist_all, acts.reshape(T * E, n).astype(np.float64),
                                                adv.reshape(-1).astype(np.float64), y.reshape(-1).astype(np.float64), 100.0)
    np.testing.assert_allclose(stats["loss"], loss, rtol=2e-4, atol=1e-6)
    shapes = NN.flat_param_shapes(S, S, 32, 32, n)
    got = NN.unflatten_params(net.get_grads().astype(np.float64), shapes)
    for name, _ in shapes:
        err = np.abs(got[name] - g[name]).max() / (np.abs(g[name]).max() + 1e-12)
        assert err < 3e-4, (name, err)


def test_trade_rollout_with_the_a3c_workers_gae():
    """gae_lambda = 0.96 (a3c/worker.py:87): the rollout's targets follow GaussianWorker.update (a3c/worker.py:232-294) --
    raw rewards, delta_t = r_t + gamma V_{t+1} - V_t, advantage discounted with gamma*lambda, target = advantage + V_t --
    applied per episode segment with bootstrap 0 behind a finished episode (done_penalty) and V(s_T) at the end."""
    from goldsrl import _ffi, _ffi_flat
    E, T, n, R = 64, 12, 2, 5
    S = 1 + 2 * n
    eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=4, n_assets=n, rnn_length=R, max_episode_steps=5)
    eng.reset()
    net = _ffi_flat.FlatNet(eng, static_size=S, temporal_size=S, rnn_length=R, num_actions=n, max_samples=E * T, scale=100.0, gae_lambda=0.96)
    _params(net, S, S, n)
    net.rollout(T); eng.wait()
    vals, rews, masks = (net.read_rollout(k, (T, E)) for k in ("values", "rewards", "masks"))
    boot, y, adv = net.read_rollout("boot", (E,)), net.read_rollout("y", (T, E)), net.read_rollout("adv", (T, E))
    assert (masks == 0).sum() == 2 * E                     # TimeLimit(5): two finished episodes inside 12 steps
    oy, oadv = np.zeros((T, E)), np.zeros((T, E))
    for b in range(E):
        ends = [t for t in range(T) if masks[t, b] == 0]
        t0 = 0
        for t1 in ends + [T - 1]:
            seg = slice(t0, t1 + 1)
            finished = t1 in ends
            bt = np.zeros(1) if finished else boot[b:b + 1].astype(np.float64)
            if t0 <= t1:
                a, tgt = O.gae(rews[seg, b:b + 1].astype(np.float64), vals[seg, b:b + 1].astype(np.float64), bt, 0.99, 0.96)
                oadv[seg, b], oy[seg, b] = a[:, 0], tgt[:, 0]
            t0 = t1 + 1
    np.testing.assert_allclose(y, oy, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv, oadv / 100.0, rtol=1e-5, atol=1e-6)
    assert np.abs(rews).max() > 0 and np.isfinite(net.train_rollout(1e-4)["loss"])
    with pytest.raises(_ffi.GrlError):
        _ffi_flat.FlatNet(eng, static_size=S, temporal_size=S, rnn_length=R, num_actions=n, gae_lambda=0.0)


def _flat_job(kind, E, T, cap, mode, monkeypatch, group=None):
    """One engine + FlatNet + three rollouts (with R6 accounting on); mode 'graph' keeps the launch-per-stage rollout; group: the
    persistent kernel's envs per workgroup (GRL_FLAT_GROUP; None = chosen by the env count)."""
    from goldsrl import _ffi
    from goldsrl import rollout as R
    if mode == "graph":
        monkeypatch.setenv("GRL_FLAT_ROLLOUT", "graph")
    else:
        monkeypatch.delenv("GRL_FLAT_ROLLOUT", raising=False)
    if group is None:
        monkeypatch.delenv("GRL_FLAT_GROUP", raising=False)
    else:
        monkeypatch.setenv("GRL_FLAT_GROUP", str(group))
    if kind == "solow":
        eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=21, max_episode_steps=cap, solow_tape_len=64)
    else:
        eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=21, n_assets=16, rnn_length=20, max_episode_steps=cap)
    eng.reset()
    eng.episodes_enable(capacity=8 * E)
    roll = R.FlatPolicyRollout(eng, T, train=False)
    out = []
    A = roll.net.cfg.num_actions
    S0 = roll.net.cfg.static_size
    for _ in range(3):
        roll.run(); eng.wait()
        d = {k: roll.net.read_rollout(k, (T, E)) for k in ("values", "rewards", "masks", "y", "adv")}
        d["actions"] = roll.net.read_rollout("actions", (T, E, A))
        d["states"] = roll.net.read_rollout("states", (T, E, S0))
        d["boot"] = roll.net.read_rollout("boot", (E,))
        if kind == "solow":
            d["histories"] = roll.net.read_rollout("histories", (T, E, 5, 2))
            for f in ("SOLOW_K", "SOLOW_Z", "SOLOW_E", "SOLOW_TAPE", "SOLOW_TAPE_POS", "NHIST", "ELAPSED", "EPISODE"):
                d["st_" + f] = eng.get_state(f)
        else:
            d["nhist"] = roll.net.read_rollout("nhist", (T, E)).view(np.int32)
            for f in ("TRADE_CASH", "TRADE_ASSETS", "TRADE_QUANTITY", "TRADE_PRICES", "NHIST", "ELAPSED", "EPISODE"):
                d["st_" + f] = eng.get_state(f)
        for o in ("obs", "obs_raw", "reward", "done"):
            d["out_" + o] = eng.read(o)
        d["done_list"] = np.sort(eng.read("done_list")[:int(eng.read("done_count")[0])])
        recs = eng.episodes_read()
        d["recs"] = np.array([(int(r["step_index"]), int(r["env"]), int(r["length"]), float(r["total_reward"])) for r in recs])
        out.append(d)
    pred = roll.net.predict_env()
    roll.net.close(); eng.close()
    return out, pred


@pytest.mark.parametrize("kind,E,cap", [("solow", 200, 7), ("solow", 4096, 1024), ("trade", 200, 9), ("trade", 1000, 1024)])
def test_persistent_rollout_is_bit_identical_to_the_graph_of_launches(kind, E, cap, monkeypatch):
    """The T-step actor loop as ONE kernel (a workgroup keeps a group of 64, 32 or 16 envs for the whole rollout: all three
    instances, and the one the env count picks) against the launch-per-stage rollout it replaces: every rollout buffer, the env
    state, the handle's outputs, the done list and the R6 records, over three consecutive rollouts with TimeLimit resets (and Solow
    tape refills) inside them; E = 200 and E = 1000 leave the last group partial at every group size."""
    T = 20
    b, pb = _flat_job(kind, E, T, cap, "graph", monkeypatch)
    for group in (64, 32, 16, None):
        a, pa = _flat_job(kind, E, T, cap, "persistent", monkeypatch, group)
        for u, (da, db) in enumerate(zip(a, b)):
            assert sorted(da) == sorted(db)
            for k in da:
                assert np.array_equal(da[k], db[k]), (kind, group, u, k)
        for k in pa:
            assert np.array_equal(pa[k], pb[k])
        if cap < 20:      # episodes ended inside the rollouts: resets, tape refills and R6 records were exercised
            assert sum(len(d["recs"]) for d in a) >= 2 * E


@pytest.mark.parametrize("kind,E,group", [("solow", 200, None), ("solow", 1100, 32), ("trade", 200, None), ("trade", 1100, 32),
                                           ("solow", 300, 64), ("trade", 300, 64)])
def test_a_rollout_that_keeps_its_activations_trains_exactly_as_one_that_recomputes_them(kind, E, group, monkeypatch):
    """grl_fnet_set_keep_activations: the persistent rollout fills the training workspace (step-major samples t * E + env, written by
    workgroups of 16 / 32 envs that share the 64-sample blocks of the layout) and the gradient step starts at the backward pass.
    Against the gradient step that runs its own forward: the local gradient of a first pass, then parameters, statistics and the next
    rollout's buffers over three updates -- bitwise (the loss statistics to 1e-6: atomics).  A keep-rollout whose parameters were replaced before training must not use
    what it kept."""
    from goldsrl import _ffi
    from goldsrl import rollout as R
    T = 6
    res = {}
    for keep in ("1", "0"):
        monkeypatch.setenv("GRL_FLAT_KEEP", keep)
        monkeypatch.delenv("GRL_FLAT_ROLLOUT", raising=False)
        if group is None:      # 16 envs per workgroup up to 4 096 envs; 32 (up to 8 192) and 64 (beyond) are forced
            monkeypatch.delenv("GRL_FLAT_GROUP", raising=False)
        else:
            monkeypatch.setenv("GRL_FLAT_GROUP", str(group))
        if kind == "solow":
            eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=5, max_episode_steps=4)
        else:
            eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=5, n_assets=16, rnn_length=20, max_episode_steps=4)
        eng.reset()
        roll = R.FlatPolicyRollout(eng, T, train=True, lr=1e-3)
        assert roll.keep_activations == (keep == "1")
        out = []
        roll.net.set_keep_activations(roll.keep_activations)
        roll.net.rollout(T); eng.wait()
        roll.net.train_rollout_grads()
        out.append(roll.net.get_grads())
        for _ in range(3):
            roll.run(); eng.wait()
            out.append(np.array([roll.last_stats[k] for k in ("loss", "policy_loss", "critic_loss_mean", "global_norm")]))
            out.append(roll.net.get_params())
            out.append(roll.net.read_rollout("values", (T, E)))
        # parameters replaced between a keep-rollout and its gradient step: the kept activations are stale
        roll.net.set_keep_activations(roll.keep_activations)
        roll.net.rollout(T); eng.wait()
        p = roll.net.get_params()
        roll.net.set_params((p * 1.01).astype(np.float32))
        roll.net.train_rollout_grads()
        out.append(roll.net.get_grads())
        res[keep] = out
        roll.net.close(); eng.close()
    assert len(res["1"]) == len(res["0"])
    for i, (x, y) in enumerate(zip(res["1"], res["0"])):
        if x.shape == (4,):      # the loss statistics: float64 atomics over the workgroups, in whatever order they finish
            np.testing.assert_allclose(x, y, rtol=1e-6)
        else:
            assert np.array_equal(x, y), (kind, E, group, i)
    assert np.abs(res["1"][0]).max() > 0


@pytest.mark.parametrize("kind", ["solow", "trade"])
def test_fast_forward_form_equals_the_layer_by_layer_form(kind, monkeypatch):
    """net_flat_fast.inc (2T + 5 stages: input halves of the GRU GEMMs hoisted out of the time loop, fused epilogues, merged
    heads) and its backward twin (net_flat_bwd_fast.inc) against the layer-by-layer kernels they replace for synthesized windows
    (GRL_FLAT_FORWARD=layers).  Forward: Solow (D = 2, even) runs the same MFMA chains, bit for bit; TradeAR1 (D = 33): the x / h
    split moves one product between two MFMA instructions -- agreement to float32 rounding.  Gradients: the fast backward sums in
    another order (16 x 16 x 4 tiles; the GRU kernels' input rows from the dz summed over time): float32 rounding."""
    from goldsrl import _ffi
    from goldsrl import rollout as R
    res = {}
    for mode in ("fast", "layers"):
        if mode == "layers":
            monkeypatch.setenv("GRL_FLAT_FORWARD", "layers")
        else:
            monkeypatch.delenv("GRL_FLAT_FORWARD", raising=False)
        monkeypatch.setenv("GRL_FLAT_ROLLOUT", "graph")      # the graph rollout goes through launch_forward in both modes
        E, T = 300, 6
        if kind == "solow":
            eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=4, max_episode_steps=4)
        else:
            eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=4, n_assets=16, rnn_length=20, max_episode_steps=4)
        eng.reset()
        roll = R.FlatPolicyRollout(eng, T, train=False)
        pred = roll.net.predict_env()
        roll.run(); eng.wait()
        vals = roll.net.read_rollout("values", (T, E))
        roll.net.train_rollout_grads()
        res[mode] = (pred, vals, roll.net.get_grads())
        roll.net.close(); eng.close()
    (pf, vf, gf), (pl, vl, gl) = res["fast"], res["layers"]
    if kind == "solow":
        for k in pf:
            assert np.array_equal(pf[k], pl[k]), k
        assert np.array_equal(vf, vl)
        np.testing.assert_allclose(gf, gl, rtol=1e-4, atol=1e-6 * np.abs(gl).max() + 1e-9)
    else:
        for k in pf:
            np.testing.assert_allclose(pf[k], pl[k], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(gf, gl, rtol=1e-4, atol=1e-6 * np.abs(gl).max() + 1e-9)
