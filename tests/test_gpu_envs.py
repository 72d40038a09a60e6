"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes -> libgoldsrl.so),
against the CPU oracle and the golden vectors captured from the reference.

Bars (BASELINE.json north_star): Swarm observation bins/positions bit-exact, Swarm float64
positions/rewards to 1e-12 relative on teacher-forced steps; Solow/Trade float32 dynamics and
returns/advantages within 1e-5 relative."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

F = None


def ffi():
    global F
    if F is None:
        from goldsrl import _ffi
        F = _ffi
    return F


def swarm_engine(E, **kw):
    f = ffi()
    return f.Engine(f.ENV_SWARM, E, **kw)


def compact_to_grid(lb, ab, G=84):
    lb = lb.astype(int); ab = ab.astype(int)
    lb[lb[:, 0] == 255] = -1
    ab[ab[:, 0] == 255] = -1
    return O.swarm_grid_from_compact(lb, ab, G)


# ------------------------------------------------------------------------------------------ Swarm
def test_swarm_step_teacher_forced_golden(golden):
    g = golden("swarm_step")
    E = len(g["x"])
    eng = swarm_engine(E, max_episode_steps=0)
    eng.set_state("SWARM_X", g["x"]); eng.set_state("SWARM_XA", g["xa"])
    eng.set_state("SWARM_PNOISE", g["particle_noise"]); eng.set_state("SWARM_ANOISE", g["agent_noise"])
    act32 = g["action"].astype(np.float32)          # the learner's shared action buffer is float32 (runners.py:9)
    eng.step(act32)
    x, xa = eng.get_state("SWARM_X"), eng.get_state("SWARM_XA")
    ox, oxa, orew, odone = O.swarm_step(g["x"], g["xa"], act32, g["agent_noise"], g["particle_noise"])
    assert np.array_equal(xa, oxa)                  # agents: only +,* in float64 -> bit-exact
    np.testing.assert_allclose(x, ox, rtol=1e-12, atol=1e-13)
    r64 = eng.read("reward_f64")
    np.testing.assert_allclose(r64, orew, rtol=1e-12)
    assert np.array_equal(eng.read("reward"), r64.astype(np.float32))
    assert not eng.read("done").any() and (r64 < 0).all()
    # and against the reference's own float64-action outputs: action rounding moves agents by <= 1e-8
    np.testing.assert_allclose(x, g["x_out"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(r64, g["reward"], rtol=1e-6)
    print("max |x - oracle| = %.3e" % np.abs(x - ox).max())


def test_swarm_step_fast_math_within_1e12(golden):
    g = golden("swarm_step")
    E = len(g["x"])
    eng = swarm_engine(E, max_episode_steps=0, flags=ffi().F_SWARM_FAST_MATH)
    eng.set_state("SWARM_X", g["x"]); eng.set_state("SWARM_XA", g["xa"])
    eng.set_state("SWARM_PNOISE", g["particle_noise"]); eng.set_state("SWARM_ANOISE", g["agent_noise"])
    act32 = g["action"].astype(np.float32)
    eng.step(act32)
    ox, oxa, orew, _ = O.swarm_step(g["x"], g["xa"], act32, g["agent_noise"], g["particle_noise"])
    np.testing.assert_allclose(eng.get_state("SWARM_X"), ox, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(eng.read("reward_f64"), orew, rtol=1e-11)


def test_swarm_reset_injected_golden(golden):
    g = golden("swarm_reset")
    seeds = (192, 7)
    st = lambda k: np.stack([g["s%d_%s" % (s, k)] for s in seeds])
    eng = swarm_engine(2)
    eng.swarm_reset_injected(st("x0"), st("xa0"), st("random_actions"), st("agent_noise"), st("particle_noise"))
    x, xa = eng.get_state("SWARM_X"), eng.get_state("SWARM_XA")
    np.testing.assert_allclose(xa, st("xa"), rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(x, st("x"), rtol=1e-11, atol=1e-12)     # 10 chained steps
    assert np.array_equal(eng.get_state("SWARM_PNOISE"), st("particle_noise")[:, 10])   # row 10 kept (quirk Q1)
    assert np.array_equal(eng.get_state("SWARM_ANOISE"), st("agent_noise")[:, 10])
    assert (eng.get_state("ELAPSED") == 0).all()
    # observation of the reset state was produced by the same launch
    for i in range(2):
        lb, ab, pos = O.swarm_observe_compact(x[i], xa[i], 84)
        assert np.array_equal(eng.read("positions")[i], pos)


def test_swarm_trajectory_timelimit_autoreset_golden(golden):
    g, r0 = golden("swarm_traj"), golden("swarm_reset")
    f = ffi()
    eng = swarm_engine(1, max_episode_steps=128, flags=f.F_RESET_FROM_SNAPSHOT)
    eng.set_state("SWARM_X", g["x0"][None]); eng.set_state("SWARM_XA", g["xa0"][None])
    eng.set_state("SWARM_PNOISE", g["particle_noise_row"][None]); eng.set_state("SWARM_ANOISE", g["agent_noise_row"][None])
    eng.set_state("RESET_X", r0["s192_x"][None]); eng.set_state("RESET_XA", r0["s192_xa"][None])
    eng.set_state("RESET_PNOISE", r0["s192_particle_noise"][10][None]); eng.set_state("RESET_ANOISE", r0["s192_agent_noise"][10][None])
    snaps = dict(zip(g["snap_steps"].tolist(), range(len(g["snap_steps"]))))
    ox, oxa = g["x0"][None].copy(), g["xa0"][None].copy()
    oa, op = g["agent_noise_row"][None], g["particle_noise_row"][None]
    for i, act in enumerate(g["actions"]):
        a32 = act.astype(np.float32)
        eng.step(a32[None])
        # oracle on the SAME float32-rounded actions: tight; reference's float64-action run: loose
        ox, oxa, orew, _ = O.swarm_step(ox, oxa, a32[None], oa, op)
        d = bool(eng.read("done")[0])
        assert d == bool(g["dones"][i]), i
        np.testing.assert_allclose(eng.read("reward_f64")[0], orew[0], rtol=1e-12)
        np.testing.assert_allclose(eng.read("reward_f64")[0], g["rewards"][i], rtol=1e-2)   # f32 vs f64 actions diverge slowly
        if d:   # auto-reset: state is the seed-192 reset state, observation is the reset one (quirk Q6)
            assert np.array_equal(eng.get_state("SWARM_X")[0], r0["s192_x"])
            assert eng.get_state("ELAPSED")[0] == 0
            assert eng.read("done_list").tolist() == [0]
            ox, oxa = r0["s192_x"][None].copy(), r0["s192_xa"][None].copy()
            oa, op = r0["s192_agent_noise"][10][None], r0["s192_particle_noise"][10][None]
            assert np.array_equal(eng.read("positions")[0], O.swarm_observe_compact(ox[0], oxa[0], 84)[2])
        gx, gxa = eng.get_state("SWARM_X"), eng.get_state("SWARM_XA")
        np.testing.assert_allclose(gx[0], ox[0], rtol=1e-12, atol=1e-13)
        assert np.array_equal(gxa, oxa)
        ox, oxa = gx, gxa     # teacher-force the oracle: the dynamics amplify 1-ulp differences ~1.15x per step
        if i in snaps:
            np.testing.assert_allclose(eng.get_state("SWARM_X")[0], g["x_snap"][snaps[i]], rtol=0, atol=5e-2)
            np.testing.assert_allclose(eng.get_state("SWARM_XA")[0], g["xa_snap"][snaps[i]], rtol=0, atol=5e-2)
    assert eng.get_state("EPISODE")[0] == 1


def test_swarm_observe_golden_bit_exact(golden):
    g = golden("swarm_obs")
    E = len(g["x"])
    eng = swarm_engine(E)
    eng.set_state("SWARM_X", g["x"]); eng.set_state("SWARM_XA", g["xa"])
    eng.observe()
    lb, ab, pos = eng.read("locust_bins"), eng.read("agent_bins"), eng.read("positions")
    assert np.array_equal(pos, g["positions"])
    for i in range(E):
        assert np.array_equal(compact_to_grid(lb[i], ab[i]), g["grid"][i]), i
    dense = eng.materialize_states()
    for i in range(E):
        ref = O.swarm_local_states(g["grid"][i], g["positions"][i]).astype(np.float32)
        assert np.array_equal(dense[i], ref), i
    assert np.array_equal(np.argwhere(dense[0][..., 2] == 1.0)[:, 1:], g["local0_onehot_idx"])


def test_swarm_batch_vs_oracle_bins_exact():
    rng = np.random.RandomState(0)
    E = 2048
    x = rng.rand(E, 80, 2) * [2.5, 1.5] + rng.normal(size=(E, 1, 2)) * [3, 0]
    x[:, :, 1] = np.abs(x[:, :, 1]) * (rng.rand(E, 80) > 0.3)        # many locusts exactly on the ground
    xa = rng.rand(E, 10, 2) * [3.5, 2.0] + x.mean(axis=1, keepdims=True) * [1, 0] - [1.0, 0.0]
    pn, an = rng.normal(size=(E, 80, 2)), rng.normal(size=(E, 10, 2))
    act = O.swarm_transform_actions(rng.normal(size=(E, 10, 2)).astype(np.float32))
    eng = swarm_engine(E, max_episode_steps=0)
    eng.set_state("SWARM_X", x); eng.set_state("SWARM_XA", xa); eng.set_state("SWARM_PNOISE", pn); eng.set_state("SWARM_ANOISE", an)
    eng.step(act)
    ox, oxa, orew, _ = O.swarm_step(x, xa, act, an, pn)
    gx, gxa = eng.get_state("SWARM_X"), eng.get_state("SWARM_XA")
    assert np.array_equal(gxa, oxa)
    np.testing.assert_allclose(gx, ox, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(eng.read("reward_f64"), orew, rtol=1e-12)
    # the observation is a pure function of the device's own positions: must equal the oracle's
    # binning of THOSE positions bit for bit, for every env
    lb, ab, pos = eng.read("locust_bins"), eng.read("agent_bins"), eng.read("positions")
    bad = 0
    for i in range(E):
        olb, oab, opos = O.swarm_observe_compact(gx[i], gxa[i], 84)
        olb = np.where(olb < 0, 255, olb).astype(np.uint8); oab = np.where(oab < 0, 255, oab).astype(np.uint8)
        bad += int(not (np.array_equal(lb[i], olb) and np.array_equal(ab[i], oab) and np.array_equal(pos[i], opos)))
    assert bad == 0
    # and the grid from the oracle's positions differs only where a point sits within 1e-12 of an edge
    mism = sum(int(not np.array_equal(compact_to_grid(lb[i], ab[i]),
                                      O.swarm_grid_from_compact(*O.swarm_observe_compact(ox[i], oxa[i], 84)[:2])))
               for i in range(0, E, 8))
    assert mism == 0


def test_swarm_device_reset_matches_oracle_generator():
    E, seed, off = 64, 1692, 1000
    eng = swarm_engine(E, seed=seed, env_id_offset=off)
    eng.reset()
    env = np.arange(E) + off
    x0 = np.stack(O.u01_pair(O.rng_block(seed, env[:, None], 0, 0, np.arange(80)[None])), axis=-1)
    xa0 = np.stack(O.u01_pair(O.rng_block(seed, env[:, None], 0, 1, np.arange(10)[None])), axis=-1)
    ra = np.stack(O.normal_pair(O.rng_block(seed, env[:, None], 0, 2, np.arange(100)[None])), axis=-1).reshape(E, 10, 10, 2)
    an = np.stack(O.normal_pair(O.rng_block(seed, env[:, None], 0, 3, np.arange(110)[None])), axis=-1).reshape(E, 11, 10, 2)
    pn = np.stack(O.normal_pair(O.rng_block(seed, env[:, None], 0, 4, np.arange(880)[None])), axis=-1).reshape(E, 11, 80, 2)
    ox, oxa = O.swarm_burn_in(x0, xa0, ra, an, pn)
    np.testing.assert_allclose(eng.get_state("SWARM_XA"), oxa, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(eng.get_state("SWARM_X"), ox, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(eng.get_state("SWARM_PNOISE"), pn[:, 10], rtol=1e-13, atol=1e-15)
    assert (eng.get_state("EPISODE") == 1).all()
    # second reset draws a different episode; RESEED mode (Swarm-eval-v0, seed=192) repeats the first
    first = eng.get_state("SWARM_X")
    eng.reset()
    assert not np.array_equal(first, eng.get_state("SWARM_X"))
    ev = swarm_engine(E, seed=seed, env_id_offset=off, flags=ffi().F_RESEED_EACH_RESET)
    ev.reset(); a = ev.get_state("SWARM_X"); ev.reset()
    assert np.array_equal(a, ev.get_state("SWARM_X")) and np.array_equal(a, first)


def test_swarm_full_size_properties():
    """BASELINE config 3 size (32 768 envs): determinism, shard invariance (RNG keyed by global env
    id), TimeLimit synchrony, rewards negative -- properties that need no CPU reference."""
    E, T = 32768, 3
    f = ffi()
    full = swarm_engine(E, seed=7)
    full.reset()
    halves = [swarm_engine(E // 2, seed=7, env_id_offset=o) for o in (0, E // 2)]
    for hgl in halves:
        hgl.reset()
    x_full = full.get_state("SWARM_X")
    assert np.array_equal(x_full, np.concatenate([hgl.get_state("SWARM_X") for hgl in halves]))
    rng = np.random.RandomState(1)
    for t in range(T):
        act = O.swarm_transform_actions(rng.normal(size=(E, 10, 2)).astype(np.float32))
        full.step(act)
        for i, hgl in enumerate(halves):
            hgl.step(act[i * E // 2:(i + 1) * E // 2])
    assert np.array_equal(full.get_state("SWARM_X"), np.concatenate([hgl.get_state("SWARM_X") for hgl in halves]))
    assert np.array_equal(full.read("positions"), np.concatenate([hgl.read("positions") for hgl in halves]))
    r = full.read("reward_f64")
    assert (r < 0).all() and np.isfinite(r).all()
    assert (full.get_state("ELAPSED") == T).all() and not full.read("done").any()
    pos = full.read("positions")
    assert pos.max() <= 83
    lb = full.read("locust_bins")
    assert ((lb == 255) | (lb < 84)).all()


def test_swarm_time_limit_all_envs_reset_together():
    E = 256
    eng = swarm_engine(E, seed=3, max_episode_steps=4)
    eng.reset()
    act = np.zeros((E, 10, 2), np.float32)
    for t in range(4):
        eng.step(act)
    assert eng.read("done").all() and int(eng.read("done_count")[0]) == E
    assert sorted(eng.read("done_list").tolist()) == list(range(E))
    assert (eng.get_state("ELAPSED") == 0).all() and (eng.get_state("EPISODE") == 2).all()
    eng.step(act)
    assert not eng.read("done").any() and (eng.get_state("ELAPSED") == 1).all()


def test_swarm_action_transform_golden(golden):
    g = golden("swarm_action")
    eng = swarm_engine(4)
    out = eng.transform_actions(g["a32"])
    np.testing.assert_allclose(out, g["out32"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(out, g["out64"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------ Solow
def solow_engine(E, **kw):
    f = ffi()
    return f.Engine(f.ENV_SOLOW, E, **kw)


@pytest.mark.parametrize("p,q", [(1, 1), (3, 2)])
def test_solow_steps_golden(golden, p, q):
    g = golden("solow")
    k_ = "p%dq%d_" % (p, q)
    T = 64
    eng = solow_engine(1, solow_p=p, solow_q=q, solow_tape_len=T, max_episode_steps=0, flags=ffi().F_RESET_FROM_SNAPSHOT)
    eng.set_state("SOLOW_Z0", g[k_ + "z0"][None])
    eng.reset()
    eng.set_state("SOLOW_TAPE", g[k_ + "tape_tail"][None])
    np.testing.assert_allclose(eng.read("obs_raw")[0], g[k_ + "obs0"], rtol=1e-6)
    for t, s in enumerate(g[k_ + "s"]):
        eng.step(np.array([[s]], np.float32))
        np.testing.assert_allclose(eng.read("obs_raw")[0], g[k_ + "obs"][t], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(eng.read("reward")[0], g[k_ + "reward"][t], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(eng.get_state("SOLOW_Z")[0], g[k_ + "z"][t], rtol=1e-5, atol=1e-6)
        assert not eng.read("done")[0]
    np.testing.assert_allclose(eng.read("obs")[0], O.solow_process_state(g[k_ + "obs"][-1]), rtol=1e-5, atol=1e-6)


def test_solow_ss_env_golden(golden):
    # `SolowSS-v0` / SolowSSEnv (fed_gym/__init__.py:15-19, fed_env.py:253-265) through the gym-style facade: reset gives
    # [k_ss(alpha), 0.] (z is NOT drawn), then the reference's own trajectory with its shock tape injected
    from goldsrl import envs
    g = golden("solow_ss")
    assert envs.registry["SolowSS-v0"][1] == int(g["max_episode_steps"]) == 1024
    env = envs.make("SolowSS-v0")
    assert isinstance(env, envs.SolowSSEnv) and (env.sigma, env.p, env.q) == (float(g["sigma"]), 1, 0)
    for _ in range(2):      # every reset puts z back to 0
        obs0 = env.reset()
        np.testing.assert_allclose(obs0, g["obs0"], rtol=1e-6)
        assert obs0[1] == 0.0
        tape = np.zeros(env._eng.cfg.solow_tape_len, np.float32)
        tape[-64:] = g["tape_tail"]
        env._eng.set_state("SOLOW_TAPE", tape[None])
        for t, s in enumerate(g["s"]):
            obs, rew, done, info = env.step(s)
            np.testing.assert_allclose(obs, g["obs"][t], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(rew, g["reward"][t], rtol=1e-5, atol=2e-6)
            assert done is False and info == {}
    # the device generator draws a sigma = 0.02 tape for it (statistics of 2048 draws)
    env.reset()
    drawn = env._eng.get_state("SOLOW_TAPE")[0]
    assert abs(drawn.std() - 0.02) < 0.002 and abs(drawn.mean()) < 0.002


def test_solow_steady_state_closed_form(golden):
    # tests/env_tests.py:145-155: sigma = 0, s = 0.1 -> k converges to (s/delta)^(1/(1-alpha))
    g = golden("solow")
    T = 2048
    eng = solow_engine(4, solow_p=1, solow_q=0, solow_sigma=0.0, solow_tape_len=T, max_episode_steps=0,
                       flags=ffi().F_RESET_FROM_SNAPSHOT)
    eng.reset()
    s = np.full((4, 1), 0.1, np.float32)
    for it in range(10000):
        if it % T == 0:
            eng.set_state("SOLOW_TAPE_POS", np.full(4, T - 1, np.int32))   # sigma=0: the tape is all zeros
        eng.step_async(s)
    eng.wait()
    k = eng.read("obs_raw")[:, 0]
    np.testing.assert_allclose(k, float(g["ss_k_closed_form"]), rtol=1e-5)
    np.testing.assert_allclose(k, float(g["ss_capital_10000"]), rtol=1e-5)


def test_solow_runner_history_and_autoreset_golden(golden):
    g = golden("solow_runner")
    steps, E = g["raw_actions"].shape[:2]
    T = g["tapes"].shape[1]
    eng = solow_engine(E, solow_tape_len=T, max_episode_steps=6, flags=ffi().F_RESET_FROM_SNAPSHOT)
    eng.set_state("SOLOW_Z0", g["z0"])
    eng.reset()
    eng.set_state("SOLOW_TAPE", g["tapes"])
    np.testing.assert_allclose(eng.read("obs"), g["init_states"], rtol=1e-6)
    for t in range(steps):
        a = eng.transform_actions(g["raw_actions"][t].astype(np.float32))        # SolowRunner: sigmoid
        np.testing.assert_allclose(a[:, 0], O.sigmoid(g["raw_actions"][t, :, 0]), rtol=1e-6)
        eng.step(a)
        np.testing.assert_allclose(eng.read("obs"), g["states"][t], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(eng.read("reward"), g["rew"][t], rtol=1e-5, atol=2e-6)
        assert np.array_equal(eng.read("done").astype(np.float32), g["done"][t])
        np.testing.assert_allclose(eng.read("history"), g["hist"][t], rtol=1e-5, atol=1e-6)
        assert np.array_equal(eng.read("history") == 0, g["hist"][t] == 0)      # zero padding pattern (Q11)


def test_solow_device_generator_and_tape_refill():
    E, T, seed = 512, 64, 99
    eng = solow_engine(E, solow_tape_len=T, max_episode_steps=5, seed=seed, env_id_offset=10)
    eng.reset()
    env = np.arange(E) + 10
    n0, n1 = O.normal_pair(O.rng_block(seed, env[:, None], 0, 9, np.arange(T // 2)[None]))
    tape = np.stack([n0, n1], axis=-1).reshape(E, T) * 0.1
    np.testing.assert_allclose(eng.get_state("SOLOW_TAPE"), tape, rtol=1e-6, atol=1e-8)
    z0 = O.normal_pair(O.rng_block(seed, env, 0, 8, 0))[0] * 0.1
    np.testing.assert_allclose(eng.get_state("SOLOW_Z")[:, 0], z0, rtol=1e-6, atol=1e-8)
    s = np.full((E, 1), 0.3, np.float32)
    for t in range(5):
        eng.step(s)
    assert eng.read("done").all() and int(eng.read("done_count")[0]) == E
    assert sorted(eng.read("done_list").tolist()) == list(range(E))         # ballot compaction lost nobody
    n0, n1 = O.normal_pair(O.rng_block(seed, env[:, None], 1, 9, np.arange(T // 2)[None]))
    np.testing.assert_allclose(eng.get_state("SOLOW_TAPE"), np.stack([n0, n1], -1).reshape(E, T) * 0.1, rtol=1e-6, atol=1e-8)
    assert (eng.get_state("SOLOW_TAPE_POS") == T - 1).all() and (eng.get_state("ELAPSED") == 0).all()
    np.testing.assert_allclose(eng.read("obs_raw")[:, 0], O.solow_k_ss(0.33), rtol=1e-6)


def test_solow_batch_vs_oracle():
    E, T = 4096, 32
    rng = np.random.RandomState(5)
    eng = solow_engine(E, solow_tape_len=T, max_episode_steps=0, flags=ffi().F_RESET_FROM_SNAPSHOT)
    eng.reset()
    k = (40 + 60 * rng.rand(E)).astype(np.float32); z = (rng.normal(size=(E, 1)) * 0.2).astype(np.float32)
    e = (rng.normal(size=(E, 1)) * 0.1).astype(np.float32); tape = (rng.normal(size=(E, T)) * 0.1).astype(np.float32)
    eng.set_state("SOLOW_K", k); eng.set_state("SOLOW_Z", z); eng.set_state("SOLOW_E", e); eng.set_state("SOLOW_TAPE", tape)
    rho_z, rho_e = O.solow_rhos(1, 1)
    ok, oz, oe = k.astype(np.float64), z.astype(np.float64), e.astype(np.float64)
    for t in range(20):
        s = rng.rand(E).astype(np.float32)
        eng.step(s[:, None])
        ok, oz, oe, oobs, orew = O.solow_step(ok, oz, oe, tape[:, T - 1 - t].astype(np.float64), s.astype(np.float64), rho_z, rho_e)
        np.testing.assert_allclose(eng.read("obs_raw"), oobs, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(eng.read("reward"), orew, rtol=1e-5, atol=5e-6)


# ------------------------------------------------------------------------------------------ TradeAR1
def trade_engine(E, **kw):
    f = ffi()
    return f.Engine(f.ENV_TRADE, E, **kw)


@pytest.mark.parametrize("n", [2, 16])
def test_trade_steps_golden(golden, n):
    g = golden("trade")
    k_ = "n%d_" % n
    eng = trade_engine(1, n_assets=n, flags=ffi().F_INJECT_NOISE)
    eng.reset()
    np.testing.assert_allclose(eng.read("obs_raw")[0], g[k_ + "obs0"])
    for t in range(len(g[k_ + "actions"])):
        eng.set_state("TRADE_NORMALS", g[k_ + "normals"][t][None])
        eng.step(g[k_ + "actions"][t][None].astype(np.float32))
        np.testing.assert_allclose(eng.read("obs_raw")[0], g[k_ + "obs"][t], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(eng.read("reward")[0], g[k_ + "reward"][t], rtol=1e-5, atol=1e-9)     # float64 account on the device
        assert bool(eng.read("done")[0]) == bool(g[k_ + "done"][t])
    if n == 2:
        np.testing.assert_allclose(eng.read("obs")[0], O.trade_process_state(eng.read("obs_raw")[0]), rtol=1e-5, atol=1e-6)


def test_trade_depletion_done_and_autoreset(golden):
    g = golden("trade")
    eng = trade_engine(1, n_assets=2, flags=ffi().F_INJECT_NOISE)
    eng.reset()
    for t in range(len(g["dep_actions"])):
        eng.set_state("TRADE_NORMALS", g["dep_normals"][t][None])
        eng.step(g["dep_actions"][t][None].astype(np.float32))
        assert bool(eng.read("done")[0]) == bool(g["dep_done"][t])
        np.testing.assert_allclose(eng.read("reward")[0], g["dep_reward"][t], rtol=1e-5, atol=1e-9)
        if not g["dep_done"][t]:
            np.testing.assert_allclose(eng.read("obs_raw")[0], g["dep_obs"][t], rtol=1e-5, atol=1e-30)
    # auto-reset: observation is the reset one (quirk Q6)
    np.testing.assert_allclose(eng.read("obs_raw")[0], [10, 0, 0, 1, 1])
    assert eng.get_state("ELAPSED")[0] == 0 and eng.read("done_list").tolist() == [0]


def test_trade_action_outside_box_is_an_error():
    f = ffi()
    eng = trade_engine(8, n_assets=2)
    eng.reset()
    a = np.zeros((8, 2), np.float32); a[3, 1] = 1.5
    eng.step_async(a)
    with pytest.raises(f.GrlError) as ei:
        eng.wait()
    assert ei.value.code == f.E_ACTION_RANGE       # reference: AssertionError (fed_env.py:301)


def test_trade_batch_vs_oracle_and_price_moments():
    E, n = 8192, 16
    rng = np.random.RandomState(2)
    eng = trade_engine(E, n_assets=n, flags=ffi().F_INJECT_NOISE, max_episode_steps=0)
    eng.reset()
    cash, assets = np.full(E, 10.0), np.full(E, 10.0); q, p = np.zeros((E, n)), np.ones((E, n))
    for t in range(12):
        act = np.tanh(rng.normal(size=(E, n))).astype(np.float32)
        nrm = rng.normal(size=(E, n)).astype(np.float32)
        eng.set_state("TRADE_NORMALS", nrm)
        eng.step(act)
        cash, assets, q, p, obs, rew, done = O.trade_step(cash, assets, q, p, act.astype(np.float64), nrm.astype(np.float64), O.trade_std_e())
        assert not done.any()
        np.testing.assert_allclose(eng.read("obs_raw"), obs, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(eng.read("reward"), rew, rtol=1e-5, atol=1e-9)
    # device generator: stationary std of log-prices stays below 2*std_p (tests/env_tests.py:115-124)
    eng2 = trade_engine(4096, n_assets=2, seed=5)
    eng2.reset()
    z = np.zeros((4096, 2), np.float32)
    ps = []
    for t in range(100):
        eng2.step(z)
        ps.append(eng2.read("obs_raw")[:, 3:5])
    assert (np.std(np.array(ps), axis=0) < 0.1).all()
    assert abs(np.log(np.array(ps)[-1]).std() - 0.05) < 0.01


# ------------------------------------------------------------------------------------------ rollout math
def test_returns_golden(golden):
    """R2 / R3 / R4 against the arrays of the reference's own train() loops (paac_loop.npz), R5 against gae_discount."""
    eng = solow_engine(1)
    p = golden("paac_loop")
    E, T, U = int(p["flat_E"]), int(p["flat_T"]), int(p["flat_updates"])
    raw = p["flat_post_rew"].reshape(U, T, E)                # the shared reward slot, before the learner's clip
    done = p["flat_post_done"].reshape(U, T, E)
    for u in range(U):       # masked + clipped (paac.py:140-172), adv / scale (paac.py:177)
        y, adv = eng.returns(raw[u], p["flat_vs"].reshape(U, T, E)[u], p["flat_boot"][u], float(p["flat_gamma"]),
                             masks=1.0 - done[u], clip=(-2.0, 2.0), scale=float(p["flat_scale"]))
        np.testing.assert_allclose(y, p["flat_y_batch"][u], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(adv, p["flat_adv_batch"][u] / float(p["flat_scale"]), rtol=1e-5, atol=2e-6)
    E, T, U = int(p["grid_E"]), int(p["grid_T"]), int(p["grid_updates"])
    for u in range(U):       # unmasked, unclipped, reward columns of quirk Q4 (paac.py:331-372)
        y, adv = eng.returns(p["grid_rewards"][u], p["grid_vs"].reshape(U, T, E * 10)[u], p["grid_boot"][u], float(p["grid_gamma"]),
                             scale=float(p["grid_scale"]))
        np.testing.assert_allclose(y, p["grid_y_batch"][u], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(adv, p["grid_adv_batch"][u] / float(p["grid_scale"]), rtol=1e-5, atol=1e-8)
    g = golden("returns")
    gamma = float(g["gamma"])
    y, adv = eng.returns(g["raw_rewards"][:, :1], g["values"][:, :1], g["boot"][:1], gamma, lam=float(g["lam"]))
    np.testing.assert_allclose(adv[:, 0], g["gae_adv"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y[:, 0], g["gae_targets"], rtol=1e-5, atol=1e-5)


def test_returns_full_size_properties():
    """Config-3 size (T=20, B=327 680): linearity in the rewards and the constant-reward closed form."""
    T, B, gamma = 20, 327680, 0.99
    rng = np.random.RandomState(0)
    eng = solow_engine(1)
    r1, r2 = rng.normal(size=(T, B)).astype(np.float32), rng.normal(size=(T, B)).astype(np.float32)
    v = np.zeros((T, B), np.float32); b0 = np.zeros(B, np.float32)
    y1, _ = eng.returns(r1, v, b0, gamma); y2, _ = eng.returns(r2, v, b0, gamma); y12, _ = eng.returns(r1 + r2, v, b0, gamma)
    np.testing.assert_allclose(y12, y1 + y2, rtol=1e-4, atol=1e-4)
    yc, advc = eng.returns(np.ones((T, B), np.float32), v + 0.5, b0 + 2.0, gamma)
    closed = np.array([(1 - gamma ** (T - t)) / (1 - gamma) + gamma ** (T - t) * 2.0 for t in range(T)])
    np.testing.assert_allclose(yc, np.broadcast_to(closed[:, None], (T, B)), rtol=1e-5)
    np.testing.assert_allclose(advc, yc - 0.5, rtol=1e-5, atol=1e-6)


def test_sigmoid_tanh_transforms_golden(golden):
    g = golden("returns")
    eng = solow_engine(1)
    x = g["sigmoid_in"].astype(np.float32)
    np.testing.assert_allclose(eng.transform_actions(x[:, None])[:, 0], g["sigmoid_out"], rtol=1e-5, atol=1e-30)
    t = golden("trade")
    tr = trade_engine(1, n_assets=1)
    np.testing.assert_allclose(tr.transform_actions(t["tanh_in"].astype(np.float32)[:, None])[:, 0], t["tanh_out"], rtol=1e-5, atol=1e-7)


def test_create_errors_are_reported_not_thrown():
    f = ffi()
    with pytest.raises(f.GrlError) as ei:
        f.Engine(f.ENV_SOLOW, 4, solow_p=0)
    assert ei.value.code == f.E_INVALID and "p=0" in str(ei.value)
    with pytest.raises(f.GrlError):
        f.Engine(f.ENV_SWARM, 4, device_id=99)
    eng = swarm_engine(4)
    with pytest.raises(ValueError):
        eng.set_state("SWARM_X", np.zeros((3, 80, 2)))
    with pytest.raises(f.GrlError) as ei:
        eng.get_state("SOLOW_K")
    assert ei.value.code == f.E_INVALID


def test_flat_envs_full_size_properties():
    """BASELINE sizes (configs 2 and 5): Solow 4 096 envs and TradeAR1-16 65 536 envs.  Size-independent properties:
    sharding invariance (env i of a 2-way split equals env i of the full batch, bit for bit: generator streams are keyed by
    the global env id), the Solow capital law k' = (1-delta) k + s y and the Trade accounting identity
    assets' = cash' + sum(q' p) at the prices the trade was made at, reward = log(assets'+1e-4) - log(assets+1e-4)."""
    f = ffi()
    # ---- Solow, config 2
    E = 4096
    full = f.Engine(f.ENV_SOLOW, E, seed=1692, max_episode_steps=1024); full.reset()
    halves = [f.Engine(f.ENV_SOLOW, E // 2, seed=1692, max_episode_steps=1024, env_id_offset=o) for o in (0, E // 2)]
    for h in halves:
        h.reset()
    rng = np.random.RandomState(0)
    for t in range(6):
        k0 = full.read("obs_raw")[:, 0].astype(np.float64)
        z0 = full.read("obs_raw")[:, 1].astype(np.float64)
        s = (1.0 / (1.0 + np.exp(-rng.normal(size=(E, 1))))).astype(np.float32)
        full.step(s)
        for h, sl in zip(halves, (slice(0, E // 2), slice(E // 2, E))):
            h.step(s[sl])
        got = np.concatenate([h.read("obs_raw") for h in halves])
        assert np.array_equal(got, full.read("obs_raw"))
        assert np.array_equal(np.concatenate([h.read("reward") for h in halves]), full.read("reward"))
        sv = np.maximum(1e-3, s[:, 0].astype(np.float64))
        y = np.exp(z0) * k0 ** 0.33
        np.testing.assert_allclose(full.read("obs_raw")[:, 0], (1 - 0.02) * k0 + sv * y, rtol=2e-5)       # fed_env.py:205-207
        np.testing.assert_allclose(full.read("reward"), np.log((1 - sv) * y + 1e-4), rtol=1e-4, atol=1e-5)
    # ---- TradeAR1-16, config 5
    E, n = 65536, 16
    full = trade_engine(E, n_assets=n, seed=77); full.reset()
    halves = [trade_engine(E // 2, n_assets=n, seed=77, env_id_offset=o) for o in (0, E // 2)]
    for h in halves:
        h.reset()
    for t in range(4):
        raw0 = full.read("obs_raw").astype(np.float64)
        assets_prev = full.get_state("TRADE_ASSETS").astype(np.float64).reshape(-1)      # valued at the previous step's prices
        act = np.tanh(rng.normal(size=(E, n))).astype(np.float32)
        full.step(act)
        for h, sl in zip(halves, (slice(0, E // 2), slice(E // 2, E))):
            h.step(act[sl])
        raw = full.read("obs_raw")
        assert np.array_equal(np.concatenate([h.read("obs_raw") for h in halves]), raw)
        alive = ~full.read("done").astype(bool)
        cash1, q1, p0 = raw[:, 0].astype(np.float64), raw[:, 1:1 + n].astype(np.float64), raw0[:, 1 + n:]
        assets1 = cash1 + (q1 * p0).sum(axis=1)          # valued at the prices the trade was made at (fed_env.py:308-311)
        rew = full.read("reward").astype(np.float64)
        np.testing.assert_allclose(full.get_state("TRADE_ASSETS").reshape(-1)[alive], assets1[alive], rtol=2e-5)
        np.testing.assert_allclose(rew[alive], (np.log(assets1 + 1e-4) - np.log(assets_prev + 1e-4))[alive], rtol=0, atol=2e-4)


# ------------------------------------------------------------------------------------------ env kwargs (round 4)
def test_swarm_step_without_wind_golden(golden):
    """SwarmEnv._step(v_action, add_wind=False) (multiagent.py:30-44) through grl_swarm_step_opts: the reference's own outputs for
    float64 and float32 action rows, teacher-forced; agents bit-exact, locusts to 1e-12."""
    g = golden("env_kwargs")
    for f32 in (False, True):
        m = g["nowind_f32"] == f32
        E = int(m.sum())
        eng = swarm_engine(E, max_episode_steps=0)
        eng.set_state("SWARM_X", g["nowind_x"][m]); eng.set_state("SWARM_XA", g["nowind_xa"][m])
        eng.set_state("SWARM_PNOISE", g["nowind_particle_noise"][m]); eng.set_state("SWARM_ANOISE", g["nowind_agent_noise"][m])
        eng.swarm_step_opts(g["nowind_action"][m].astype(np.float32 if f32 else np.float64), add_wind=False)
        assert np.array_equal(eng.get_state("SWARM_XA"), g["nowind_xa_out"][m])
        np.testing.assert_allclose(eng.get_state("SWARM_X"), g["nowind_x_out"][m], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(eng.read("reward_f64"), g["nowind_reward"][m], rtol=1e-12)
        eng.close()
    # the facade's keyword (goldsrl/envs/multiagent.py) reaches the same entry point
    from goldsrl.envs.multiagent import SwarmEnv
    env = SwarmEnv(seed=192)
    env.reset()
    env._eng.set_state("SWARM_X", g["nowind_x"][:1]); env._eng.set_state("SWARM_XA", g["nowind_xa"][:1])
    env._eng.set_state("SWARM_PNOISE", g["nowind_particle_noise"][:1]); env._eng.set_state("SWARM_ANOISE", g["nowind_agent_noise"][:1])
    (x, xa), r, d, info = env._step(g["nowind_action"][0], add_wind=False)
    assert np.array_equal(xa, g["nowind_xa_out"][0]) and not d and info == {}
    np.testing.assert_allclose(r, g["nowind_reward"][0], rtol=1e-12)


def test_trade_starting_balance_std_p_n_assets_golden(golden):
    """TradeAR1Env(starting_balance=25, n_assets=3, std_p=0.1) and the auto-reset of a depleted starting_balance=3 env
    (fed_env.py:269-330) against the reference's own episode."""
    g = golden("env_kwargs")
    f = ffi()
    n, sb, sp = int(g["tk_n"]), float(g["tk_starting_balance"]), float(g["tk_std_p"])
    eng = trade_engine(1, n_assets=n, trade_std_p=sp, trade_starting_balance=sb, flags=f.F_INJECT_NOISE)
    eng.reset()
    np.testing.assert_allclose(eng.read("obs_raw")[0], g["tk_obs0"])
    for t in range(len(g["tk_actions"])):
        eng.set_state("TRADE_NORMALS", g["tk_normals"][t][None])
        eng.step(g["tk_actions"][t][None].astype(np.float32))
        np.testing.assert_allclose(eng.read("obs_raw")[0], g["tk_obs"][t], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(eng.read("reward")[0], g["tk_reward"][t], rtol=1e-5, atol=1e-9)
        assert bool(eng.read("done")[0]) == bool(g["tk_done"][t])
    eng.close()
    eng = trade_engine(1, n_assets=2, trade_starting_balance=3.0, flags=f.F_INJECT_NOISE)
    eng.reset()
    np.testing.assert_allclose(eng.read("obs_raw")[0], g["td_obs0"])
    for t in range(len(g["td_actions"])):
        eng.set_state("TRADE_NORMALS", g["td_normals"][t][None])
        eng.step(g["td_actions"][t][None].astype(np.float32))
        assert bool(eng.read("done")[0]) == bool(g["td_done"][t])
        np.testing.assert_allclose(eng.read("reward")[0], g["td_reward"][t], rtol=1e-5, atol=1e-9)
        if not g["td_done"][t]:
            np.testing.assert_allclose(eng.read("obs_raw")[0], g["td_obs"][t], rtol=1e-5, atol=1e-30)
    np.testing.assert_allclose(eng.read("obs_raw")[0], g["td_reset_obs"])      # auto-reset: the starting balance again (quirk Q6)
    np.testing.assert_allclose(eng.get_state("TRADE_CASH"), [3.0]); np.testing.assert_allclose(eng.get_state("TRADE_ASSETS"), [3.0])
    eng.close()
    # the facade's constructor arguments, and a starting balance the log cannot take
    from goldsrl.envs.fed_env import TradeAR1Env
    env = TradeAR1Env(starting_balance=sb, n_assets=n, std_p=sp)
    np.testing.assert_allclose(env.reset(), g["tk_obs0"])
    assert env.starting_balance == sb and env.std_e == float(g["tk_std_e"])
    with pytest.raises(f.GrlError):
        trade_engine(1, n_assets=2, trade_starting_balance=0.0)
