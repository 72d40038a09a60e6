"""Pin the numpy oracle (oracle/oracle.py) against golden vectors captured from the
unmodified reference (tests/golden/gen_golden.py).  CPU only."""
import numpy as np

from oracle import oracle as O


def test_swarm_step_teacher_forced(golden):
    g = golden("swarm_step")
    x, xa, r, d = O.swarm_step(g["x"], g["xa"], g["action"], g["agent_noise"], g["particle_noise"])
    # same numpy, same association order -> bit-exact against the reference
    assert np.array_equal(xa, g["xa_out"])
    assert np.array_equal(x, g["x_out"])
    assert np.array_equal(r, g["reward"])
    assert np.array_equal(d, g["done"])
    assert (r < 0).all() and not d.any()          # tests/env_tests.py:24-25


def test_swarm_reset_burn_in(golden):
    g = golden("swarm_reset")
    for seed in (192, 7):
        k = "s%d_" % seed
        x, xa = O.swarm_burn_in(g[k + "x0"][None], g[k + "xa0"][None], g[k + "random_actions"][None],
                                g[k + "agent_noise"][None], g[k + "particle_noise"][None])
        assert np.array_equal(x[0], g[k + "x"]) and np.array_equal(xa[0], g[k + "xa"])


def test_swarm_trajectory_time_limit_and_autoreset(golden):
    g = golden("swarm_traj")
    r0 = golden("swarm_reset")
    x, xa = g["x0"][None].copy(), g["xa0"][None].copy()
    a_row, p_row = g["agent_noise_row"][None], g["particle_noise_row"][None]
    elapsed, snaps = 0, dict(zip(g["snap_steps"].tolist(), range(len(g["snap_steps"]))))
    for i, act in enumerate(g["actions"]):
        x, xa, r, d = O.swarm_step(x, xa, act[None], a_row, p_row)
        elapsed += 1
        done = bool(d[0]) or O.time_limit_done(elapsed, 128)
        assert r[0] == g["rewards"][i] and done == bool(g["dones"][i])
        if done:   # Swarm-eval-v0 reseeds with 192 on every reset -> the seed-192 reset fixture
            x, xa = r0["s192_x"][None].copy(), r0["s192_xa"][None].copy()
            a_row, p_row = r0["s192_agent_noise"][10][None], r0["s192_particle_noise"][10][None]
            elapsed = 0
        if i in snaps:
            assert np.array_equal(x[0], g["x_snap"][snaps[i]])
            assert np.array_equal(xa[0], g["xa_snap"][snaps[i]])
    assert g["dones"].sum() == 1 and g["dones"][127]


def test_swarm_observe(golden):
    g = golden("swarm_obs")
    assert bool(g["on_edge_case_exact"])
    for i in range(len(g["x"])):
        lb, ab, pos = O.swarm_observe_compact(g["x"][i], g["xa"][i], 84)
        grid = O.swarm_grid_from_compact(lb, ab, 84)
        assert np.array_equal(grid, g["grid"][i]), i
        assert np.array_equal(pos, g["positions"][i]), i
    lb, ab, pos = O.swarm_observe_compact(g["x"][0], g["xa"][0], 84)
    loc = O.swarm_local_states(O.swarm_grid_from_compact(lb, ab), pos)
    idx = np.array([np.argwhere(l[:, :, 2] == 1.0)[0] for l in loc])
    assert np.array_equal(idx, g["local0_onehot_idx"])
    # quirk Q2: one-hot is offset (+1,+1) from the density bin unless clamped
    inside = (ab[:, 0] >= 0) & (ab[:, 0] < 83) & (ab[:, 1] < 83)
    assert np.array_equal(pos[inside].astype(int), ab[inside] + 1)


def test_swarm_action_transform(golden):
    g = golden("swarm_action")
    assert np.array_equal(O.swarm_transform_actions(g["a64"]), g["out64"])
    out32 = O.swarm_transform_actions(g["a32"])
    assert out32.dtype == np.float32
    np.testing.assert_allclose(out32, g["out32"], rtol=2e-7, atol=0)


def test_solow(golden):
    g = golden("solow")
    assert O.solow_k_ss(0.33) == g["k_ss_033"]
    for (p, q) in ((1, 1), (3, 2)):
        k_ = "p%dq%d_" % (p, q)
        rho_z, rho_e = O.solow_rhos(p, q)
        assert np.array_equal(rho_z, g[k_ + "rho_z"]) and np.array_equal(rho_e, g[k_ + "rho_e"])
        k = np.array([O.solow_k_ss(0.33)])
        z, e = g[k_ + "z0"][None].copy(), g[k_ + "e0"][None].copy()
        assert np.array_equal(np.array([k[0], z[0, -1]]), g[k_ + "obs0"])
        tape = g[k_ + "tape_tail"]
        for t, s in enumerate(g[k_ + "s"]):
            e_t = tape[-1 - t]          # es.pop(): consumed back to front (quirk Q8)
            k, z, e, obs, rew = O.solow_step(k, z, e, np.array([e_t]), np.array([s]), rho_z, rho_e)
            np.testing.assert_allclose(obs[0], g[k_ + "obs"][t], rtol=1e-15, atol=0)
            np.testing.assert_allclose(rew[0], g[k_ + "reward"][t], rtol=1e-14, atol=1e-16)
            np.testing.assert_allclose(z[0], g[k_ + "z"][t], rtol=1e-15)
    # analytical steady state (tests/env_tests.py:145-155)
    k = np.array([float(g["ss_k0"])]); z = np.zeros((1, 1)); e = np.zeros((1, 1))
    rho_z, _ = O.solow_rhos(1, 0)
    for _ in range(10000):
        k, z, e, obs, rew = O.solow_step(k, z, e, np.zeros(1), np.array([0.1]), rho_z, np.array([0.5]))
    np.testing.assert_almost_equal(k[0], (0.1 / 0.02) ** (1 / (1 - 0.33)))
    np.testing.assert_allclose(k[0], g["ss_capital_10000"], rtol=1e-13)
    assert np.array_equal(O.solow_process_state(g["proc_in"]), g["proc_out"])


def test_solow_ss(golden):
    # SolowSSEnv (fed_env.py:253-265): p=1, q=0, sigma=0.02; reset z = 0, e = 0, k = k_ss(alpha)
    g = golden("solow_ss")
    rho_z, rho_e = O.solow_rhos(1, 0)
    assert np.array_equal(rho_z, g["rho_z"]) and float(g["rho_e"]) == 0.5 and int(g["max_episode_steps"]) == 1024
    k, z, e = np.array([O.solow_k_ss(0.33)]), np.zeros((1, 1)), np.zeros((1, 1))
    assert np.array_equal(g["obs0"], [k[0], 0.0])
    tape = list(g["tape_tail"])
    for t, s in enumerate(g["s"]):
        k, z, e_, obs, rew = O.solow_step(k, z, e, np.array([tape.pop()]), np.array([s]), rho_z, rho_e)
        # q = 0: the MA window is the scalar e_t of the previous step (fed_env.py:224-227 keeps self.e = [e_t])
        e = e_[:, -1:]
        np.testing.assert_allclose(obs[0], g["obs"][t], rtol=1e-15, atol=0)
        np.testing.assert_allclose(rew[0], g["reward"][t], rtol=1e-14, atol=1e-16)      # numpy's scalar vs array log differ by an ulp


def test_solow_runner_history_and_autoreset(golden):
    g = golden("solow_runner")
    steps, E = g["raw_actions"].shape[:2]
    rho_z, rho_e = O.solow_rhos(1, 1)
    k = np.full(E, O.solow_k_ss(0.33)); z = g["z0"].copy(); e = np.zeros((E, 1))
    tape_pos = np.full(E, g["tapes"].shape[1] - 1)
    elapsed = np.zeros(E, int); n_hist = np.zeros(E, int)   # worker list starts empty (emulator_runner.py:23)
    for t in range(steps):
        s = O.sigmoid(g["raw_actions"][t, :, 0]).astype(np.float32)
        e_t = g["tapes"][np.arange(E), tape_pos]; tape_pos -= 1
        k, z, e, obs, rew = O.solow_step(k, z, e, e_t, s.astype(np.float64), rho_z, rho_e)
        elapsed += 1
        done = O.time_limit_done(elapsed, 6)
        # Q6: terminal reward is reported, observation is the reset one
        k = np.where(done, O.solow_k_ss(0.33), k)
        z = np.where(done[:, None], g["z0"], z); e = np.where(done[:, None], 0.0, e)
        tape_pos = np.where(done, g["tapes"].shape[1] - 1, tape_pos)
        elapsed = np.where(done, 0, elapsed)
        n_hist = np.where(done, 1, n_hist + 1)
        state = O.solow_process_state(np.stack([k, z[:, -1]], axis=1))
        np.testing.assert_allclose(state, g["states"][t], rtol=1e-6)   # (1-s) is float32 in the reference
        np.testing.assert_allclose(rew, g["rew"][t], rtol=2e-6, atol=1e-7)
        assert np.array_equal(done.astype(np.float32), g["done"][t])
        hist = O.history_window(g["states"][t], n_hist, 5)
        assert np.array_equal(hist, g["hist"][t])          # quirk Q11


def test_trade(golden):
    g = golden("trade")
    assert O.trade_std_e() == g["n2_std_e"]
    for n in (2, 16):
        k_ = "n%d_" % n
        cash, assets = np.array([10.0]), np.array([10.0])
        q, p = np.zeros((1, n)), np.ones((1, n))
        assert np.array_equal(np.concatenate([cash, q[0], p[0]]), g[k_ + "obs0"])
        for t in range(len(g[k_ + "actions"])):
            cash, assets, q, p, obs, rew, done = O.trade_step(
                cash, assets, q, p, g[k_ + "actions"][t][None], g[k_ + "normals"][t][None], O.trade_std_e())
            np.testing.assert_allclose(obs[0], g[k_ + "obs"][t], rtol=1e-14, atol=1e-15)
            np.testing.assert_allclose(rew[0], g[k_ + "reward"][t], rtol=1e-12, atol=1e-15)
            assert bool(done[0]) == bool(g[k_ + "done"][t])
    cash, assets = np.array([10.0]), np.array([10.0]); q, p = np.zeros((1, 2)), np.ones((1, 2))
    for t in range(len(g["dep_actions"])):
        cash, assets, q, p, obs, rew, done = O.trade_step(
            cash, assets, q, p, g["dep_actions"][t][None], g["dep_normals"][t][None], O.trade_std_e())
        np.testing.assert_allclose(obs[0], g["dep_obs"][t], rtol=1e-13, atol=1e-300)
        assert bool(done[0]) == bool(g["dep_done"][t])
    assert g["dep_done"][-1]
    np.testing.assert_allclose(O.trade_process_state(g["proc_in"]), g["proc_out"], rtol=1e-15)
    np.testing.assert_allclose(np.tanh(g["tanh_in"]), g["tanh_out"], rtol=1e-15)


def test_returns_and_misc(golden):
    g = golden("returns")
    gamma, lam = float(g["gamma"]), float(g["lam"])
    clipped = O.rescale_reward(g["raw_rewards"])
    assert np.array_equal(clipped.astype(np.float64), g["clipped_rewards"])
    # the n-step return loops are pinned by the reference's own train() loops: tests/test_oracle_learner_golden.py
    y_u, adv_u = O.nstep_returns(g["raw_rewards"].astype(np.float64), g["values"], g["boot"], gamma)
    a, tgt = O.gae(g["raw_rewards"][:, :1].astype(np.float64), g["values"][:, :1], g["boot"][:1], gamma, lam)
    np.testing.assert_allclose(a[:, 0], g["gae_adv"], rtol=1e-13)
    np.testing.assert_allclose(tgt[:, 0], g["gae_targets"], rtol=1e-13)
    # lambda = 1 telescopes to the PAAC n-step advantage
    a1, t1 = O.gae(g["raw_rewards"].astype(np.float64), g["values"], g["boot"], gamma, 1.0)
    np.testing.assert_allclose(a1, adv_u, rtol=1e-6, atol=1e-6)   # f32 gamma*boot product in the reference
    for s, lr in zip(g["lr_steps"], g["lrs"]):
        assert O.get_lr(int(s), 1e-4, 80000000) == lr
    np.testing.assert_allclose(O.sigmoid(g["sigmoid_in"]), g["sigmoid_out"], rtol=1e-15, atol=0)


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    z = O.philox4x32(np.zeros(4, np.uint32), np.zeros(2, np.uint32))
    assert [hex(v) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = np.full(4, 0xFFFFFFFF, np.uint32)
    z = O.philox4x32(f, f[:2])
    assert [hex(v) for v in z] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    z = O.philox4x32(np.array([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], np.uint32),
                     np.array([0xa4093822, 0x299f31d0], np.uint32))
    assert [hex(v) for v in z] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    u0, u1 = O.u01_pair(O.rng_block(1692, np.arange(1000), 0, 0, 0))
    assert (u0 >= 0).all() and (u0 < 1).all() and abs(u0.mean() - 0.5) < 0.05
    n0, n1 = O.normal_pair(O.rng_block(1692, np.arange(20000), 0, 1, 3))
    assert abs(n0.mean()) < 0.03 and abs(n1.std() - 1) < 0.03


def test_episode_bookkeeping_restatement_on_a_hand_example():
    # paac.py:142-157 / 331-349 by hand: 2 envs, 3 steps; env 1 finishes at step 1 (0-based), env 0 at step 2
    rew = np.array([[1.0, 10.0], [2.0, 20.0], [4.0, 40.0]], np.float32)
    done = np.array([[0, 0], [0, 1], [1, 0]], np.uint8)
    recs, total, steps, gs = O.episode_bookkeeping(rew, done, global_step=100)
    # global_step += 1 per env inside a step, in env order: (t=1, e=1) is the 4th increment, (t=2, e=0) the 5th
    assert recs == [(104, 1, 2, 30.0), (105, 0, 3, 7.0)]
    assert total.tolist() == [0.0, 40.0] and steps.tolist() == [0, 1] and gs == 106
    # carried over into the next rollout
    recs2, total2, steps2, gs2 = O.episode_bookkeeping(rew[:1], np.array([[0, 1]], np.uint8), total, steps, gs)
    assert recs2 == [(108, 1, 2, 50.0)] and total2.tolist() == [1.0, 0.0] and gs2 == 108
    # float64 accumulation of the float32 rewards (numpy 1.13: `0 + np.float32` is float64)
    r = np.full((3, 1), np.float32(0.1))
    recs3, *_ = O.episode_bookkeeping(r, np.array([[0], [0], [1]], np.uint8))
    assert recs3[0][3] == float(np.float32(0.1)) * 3 or abs(recs3[0][3] - 3 * float(np.float32(0.1))) < 1e-15


# ------------------------------------------------------------------------------------------ env kwargs (round 4)
def test_swarm_step_without_wind_golden(golden):
    """SwarmEnv._step(v_action, add_wind=False) (multiagent.py:30-44): float64 and float32 action rows, teacher-forced."""
    g = golden("env_kwargs")
    for f32 in (False, True):
        m = g["nowind_f32"] == f32
        act = g["nowind_action"][m].astype(np.float32 if f32 else np.float64)
        x, xa, rew, done = O.swarm_step(g["nowind_x"][m], g["nowind_xa"][m], act, g["nowind_agent_noise"][m], g["nowind_particle_noise"][m],
                                        add_wind=False)
        assert np.array_equal(xa, g["nowind_xa_out"][m])
        np.testing.assert_allclose(x, g["nowind_x_out"][m], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(rew, g["nowind_reward"][m], rtol=1e-13)
        # and the wind does matter: the default step moves the agents elsewhere
        _, xa_w, _, _ = O.swarm_step(g["nowind_x"][m], g["nowind_xa"][m], act, g["nowind_agent_noise"][m], g["nowind_particle_noise"][m])
        assert np.abs(xa_w - xa).max() > 0.04


def test_trade_starting_balance_std_p_n_assets_golden(golden):
    """TradeAR1Env(starting_balance=25, n_assets=3, std_p=0.1) (fed_env.py:269-330)."""
    g = golden("env_kwargs")
    n, sb = int(g["tk_n"]), float(g["tk_starting_balance"])
    std_e = O.trade_std_e(float(g["tk_std_p"]))
    assert std_e == float(g["tk_std_e"])
    assert np.array_equal(g["tk_obs0"], np.concatenate([[sb], np.zeros(n), np.ones(n)]))
    cash, assets, q, p = np.array([sb]), np.array([sb]), np.zeros((1, n)), np.ones((1, n))
    for t in range(len(g["tk_actions"])):
        cash, assets, q, p, obs, rew, done = O.trade_step(cash, assets, q, p, g["tk_actions"][t][None], g["tk_normals"][t][None], std_e)
        np.testing.assert_allclose(obs[0], g["tk_obs"][t], rtol=1e-13)
        np.testing.assert_allclose(rew[0], g["tk_reward"][t], rtol=1e-10, atol=1e-15)
        assert bool(done[0]) == bool(g["tk_done"][t])
    assert np.array_equal(g["td_obs0"], [3.0, 0, 0, 1, 1]) and np.array_equal(g["td_reset_obs"], g["td_obs0"]) and g["td_done"][-1]
