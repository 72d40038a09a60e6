"""CPU: the oracle's restatement of the LEARNER LOOP rows (R2, R3, R4, R6, R7, quirk Q4, S9) against fixtures captured from the
reference's own PAACLearner.train / GridPAACLearner.train / SwarmRunner._run (tests/golden/gen_golden_learner.py: the
unmodified loops under a canned network).  Nothing here touches the GPU."""
import numpy as np

from oracle import oracle as O


def _dense(idx, val, shape):
    a = np.zeros(tuple(int(s) for s in shape))
    a[tuple(idx.T)] = val
    return a


def test_flat_loop_masked_clipped_returns_and_feed_order(golden):
    """paac.py:140-187: masks = 1 - done, rewards clipped to +-2, R <- r + gamma R m, y = R, adv = R - V; feeds are the
    time-major flattening, advantages divided by network.scale."""
    g = golden("paac_loop")
    E, T, U = int(g["flat_E"]), int(g["flat_T"]), int(g["flat_updates"])
    gamma, scale = float(g["flat_gamma"]), float(g["flat_scale"])
    post_rew = g["flat_post_rew"].reshape(U, T, E)
    post_done = g["flat_post_done"].reshape(U, T, E)
    vs = g["flat_vs"].reshape(U, T, E)
    for u in range(U):
        # what the loop stored (its own locals) is what the oracle derives from the shared slots
        assert np.array_equal(g["flat_rewards"][u], O.rescale_reward(post_rew[u]).astype(np.float64))
        assert np.array_equal(g["flat_episodes_over_masks"][u], (1.0 - post_done[u]).astype(np.float64))
        assert np.array_equal(g["flat_values"][u], vs[u].astype(np.float64))
        y, adv = O.nstep_returns(g["flat_rewards"][u], g["flat_values"][u], g["flat_boot"][u], gamma,
                                 g["flat_episodes_over_masks"][u])
        assert np.array_equal(y, g["flat_y_batch"][u]) and np.array_equal(adv, g["flat_adv_batch"][u])
        assert np.array_equal(g["flat_feed_critic_target"][u], y.reshape(-1))                 # R4: index t*E + e
        assert np.array_equal(g["flat_feed_advantages"][u], adv.reshape(-1) / scale)
        assert g["flat_lr"][u] == O.get_lr(int(g["flat_global_step"][u]), float(g["flat_lr0"]), int(g["flat_anneal"]))   # R7
        assert g["flat_feed_lr"][u] == g["flat_lr"][u]
    assert (post_done.sum(axis=(1, 2)) > 0).any() and (np.abs(post_rew) > 2).sum() >= 0
    # actions fed to the loss are the RAW samples mu + sigma * eps (paac.py:36,130,178), time-major
    assert np.array_equal(g["flat_feed0_actions"], g["flat_actions"][0].reshape(T * E, 1))
    # the env sees sigmoid(raw) through the float32 shared array (paac.py:126-128; emulator_runner.py:77-79)
    act = g["flat_actions"].reshape(U * T, E, 1)
    assert np.array_equal(g["flat_post_act"], O.sigmoid(act).astype(np.float32))
    # states / histories fed = the shared slots at the time of each forward pass, time-major
    keep = g["flat_shared_states"].shape[1]
    fs = g["flat_feed0_states"].reshape(T, E, 2)
    fh = g["flat_feed0_history"].reshape(T, E, 5, 2)
    for t in range(T):
        assert np.array_equal(fs[t, :keep], g["flat_shared_states"][t]) and np.array_equal(fh[t, :keep], g["flat_shared_hist"][t])
    assert list(g["flat_merged_summary_steps"]) == list(range(1, U + 1))


def test_flat_loop_env_slots_replay_on_the_oracle(golden):
    """The Solow worker path behind the loop (emulator_runner.py:38-79 in two worker processes over np.split shards): states,
    rewards, dones, history windows of every step replayed by the oracle's Solow step + TimeLimit(6) + seeded auto-reset."""
    g = golden("paac_loop")
    E, T, U, cap = int(g["flat_E"]), int(g["flat_T"]), int(g["flat_updates"]), int(g["flat_cap"])
    rho_z, rho_e = O.solow_rhos(1, 1)
    tape = g["flat_tape"]
    k = np.full(E, float(g["flat_k0"])); z = np.tile(g["flat_z0"], (E, 1)); e = np.tile(g["flat_e0"], (E, 1))
    pos = np.full(E, len(tape) - 1); elapsed = np.zeros(E, int); nh = np.zeros(E, int)
    keep = g["flat_shared_states"].shape[1]
    for i in range(U * T):
        a = g["flat_post_act"][i][:, 0].astype(np.float64)
        k, z, e, obs, rew = O.solow_step(k, z, e, tape[pos], a, rho_z, rho_e)
        pos -= 1; elapsed += 1
        done = O.time_limit_done(elapsed, cap)
        assert np.array_equal(done.astype(np.float32), g["flat_post_done"][i])
        np.testing.assert_allclose(rew.astype(np.float32), g["flat_post_rew"][i], rtol=1e-6)
        # Q6: a finished env reports the terminal reward and the RESET observation; every reset replays the seeded episode
        k = np.where(done, float(g["flat_k0"]), k); z[done] = g["flat_z0"]; e[done] = g["flat_e0"]
        pos[done] = len(tape) - 1; elapsed[done] = 0
        raw = np.where(done[:, None], np.stack([k, z[:, -1]], 1), obs)
        st = O.solow_process_state(raw)
        np.testing.assert_allclose(st, g["flat_post_states"][i], rtol=1e-13)
        nh = np.where(done, 1, nh + 1)
        if i + 1 < U * T:
            hist = O.history_window(st, nh, 5)
            # the learner's history slot is a float32 shared array (pad_sequences(dtype='float32'), paac.py:88-90; Q7)
            np.testing.assert_allclose(hist[:keep].astype(np.float32), g["flat_shared_hist"][i + 1], rtol=1e-7)
    assert g["flat_post_done"].sum() == 2 * E           # steps 6 and 12 of 12


def test_flat_loop_bookkeeping(golden):
    """R6 (paac.py:142-157): rl/reward points, their global_step, total_rewards and the running sums, from the reference's own
    loop.  The reference ran under numpy 2 here, where `0 + np.float32` stays float32 (under its numpy 1.13 pin the sum is
    float64, which is what the oracle and the device do): totals agree to float32 resolution, everything else exactly."""
    g = golden("paac_loop")
    E = int(g["flat_E"])
    recs, total, steps, gs = O.episode_bookkeeping(g["flat_post_rew"], g["flat_post_done"])
    assert str(g["flat_rl_value_type"][0]) == "float32"
    assert [r[0] for r in recs] == list(g["flat_rl_step"]) and [r[1] for r in recs] == list(g["flat_rl_env"])
    assert all(r[2] == 6 for r in recs)
    np.testing.assert_allclose([r[3] for r in recs], g["flat_rl_reward"], rtol=2e-6, atol=1e-6)      # float32 running sums in the capture
    np.testing.assert_allclose([r[3] / r[2] for r in recs], g["flat_total_rewards_final"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(total, g["flat_running_total"], rtol=2e-6, atol=1e-7)
    assert np.array_equal(steps, g["flat_running_steps"]) and gs == int(g["flat_global_step"][-1]) == int(g["flat_log_steps"])
    # global_step of a record = steps before it in (t, env) order (paac.py:149): record at loop step i, env e -> i*E + e + 1
    t_abs = (g["flat_rl_step"] - 1) // E
    assert np.array_equal((g["flat_rl_step"] - 1) % E, g["flat_rl_env"]) and set(t_abs) == {5, 11}
    np.testing.assert_allclose(float(g["flat_log_last_ten"]), np.mean([r[3] / r[2] for r in recs][-10:]), rtol=2e-6, atol=1e-6)


def test_grid_loop_unmasked_returns_and_reward_columns(golden):
    """paac.py:331-372: rewards[t, e_idx] = reward of env e_idx for e_idx < E only (quirk Q4: E of the E*10 columns), no
    clipping, no done mask (Q5), bootstrap from the reset observation's value."""
    g = golden("paac_loop")
    E, T, U = int(g["grid_E"]), int(g["grid_T"]), int(g["grid_updates"])
    B = E * 10
    gamma, scale = float(g["grid_gamma"]), float(g["grid_scale"])
    post_rew = g["grid_shared_rew"][1:].reshape(U, T, E, 10)
    post_done = g["grid_shared_done"][1:].reshape(U, T, E, 10)
    assert (post_rew == post_rew[..., :1]).all() and (post_done == post_done[..., :1]).all()     # scalar broadcast over agents
    for u in range(U):
        rw = g["grid_rewards"][u]
        assert rw.shape == (T, B) and (rw[:, E:] == 0).all()
        assert np.array_equal(rw[:, :E], post_rew[u, :, :, 0].astype(np.float64))
        assert np.array_equal(g["grid_values"][u], g["grid_vs"].reshape(U, T, B)[u].astype(np.float64))
        y, adv = O.nstep_returns(rw, g["grid_values"][u], g["grid_boot"][u], gamma)
        assert np.array_equal(y, g["grid_y_batch"][u]) and np.array_equal(adv, g["grid_adv_batch"][u])
        assert np.array_equal(g["grid_feed_critic_target"][u], y.reshape(-1))
        assert np.array_equal(g["grid_feed_advantages"][u], adv.reshape(-1) / scale)
        assert np.array_equal(g["grid_feed_actions"][u], g["grid_actions"][u].reshape(T * B, 2))
    assert post_done[..., 0].sum() == E                 # TimeLimit(4): every env finishes once in 6 steps
    assert sorted(str(k) for k in g["grid_feed_keys"]) == ["actions", "advantages", "critic_target", "learning_rate", "states"]
    # Q7: dtypes of the shared slots (runners.py:9: uint8 -> c_uint)
    assert [str(d) for d in g["grid_shared_dtypes"]] == ["float64", "float32", "uint32", "float32", "float32", "float32"]
    # feed order of the states: time-major, env-major inside a step, agent fastest (paac.py:319,367)
    st = _dense(g["grid_shared_states_idx"], g["grid_shared_states_val"], g["grid_shared_states_shape"])
    feed = _dense(g["grid_feed0_states_idx"], g["grid_feed0_states_val"], g["grid_feed0_states_shape"])
    assert np.array_equal(feed.reshape(T, E, 10, 84, 84, 3), st[:T])
    # R6 on the grid loop: column 0 of the env's row
    recs, total, steps, gs = O.episode_bookkeeping(post_rew.reshape(U * T, E, 10)[:, :, 0], post_done.reshape(U * T, E, 10)[:, :, 0])
    assert [r[0] for r in recs] == list(g["grid_rl_step"]) and [r[1] for r in recs] == list(g["grid_rl_env"])
    np.testing.assert_allclose([r[3] for r in recs], g["grid_rl_reward"], rtol=2e-6)
    np.testing.assert_allclose([r[3] / r[2] for r in recs], g["grid_total_rewards_final"], rtol=2e-6)
    np.testing.assert_allclose(total, g["grid_running_total"], rtol=2e-6)
    assert np.array_equal(steps, g["grid_running_steps"]) and gs == int(g["grid_global_step"][-1])


def _replay_swarm_worker(golden, raw_actions_f32, cap, n_env, init_check=None):
    """Oracle replay of SwarmRunner._run's per-env body over the seed-192 env of swarm_reset.npz."""
    r = golden("swarm_reset")
    x0, xa0 = r["s192_x"], r["s192_xa"]
    an, pn = r["s192_agent_noise"][10], r["s192_particle_noise"][10]
    x = np.tile(x0, (n_env, 1, 1)); xa = np.tile(xa0, (n_env, 1, 1))
    elapsed = np.zeros(n_env, int)
    out = []
    for act in raw_actions_f32:
        x, xa, rew, d = O.swarm_step(x, xa, act, np.tile(an, (n_env, 1, 1)), np.tile(pn, (n_env, 1, 1)))
        elapsed += 1
        done = d | O.time_limit_done(elapsed, cap)
        x[done], xa[done] = x0, xa0
        elapsed[done] = 0
        obs = [O.swarm_observe_compact(x[i], xa[i], 84) for i in range(n_env)]
        out.append((x.copy(), xa.copy(), rew, done, obs))
    return out


def test_swarm_runner_slots(golden):
    """S9 from SwarmRunner._run itself (emulator_runner.py:120-151), rnn_length 1 and 2: step with the float32 action row,
    TimeLimit(4), reset observation on done (Q6), positions, scalar reward/done broadcast, and the HISTORY slot."""
    g = golden("swarm_runner")
    for rnn in (1, 2):
        k = "r%d_" % rnn
        acts = g[k + "act"]                                    # (steps, E, 10, 2) float32: what the worker read
        raw = g[k + "raw_actions"]
        assert np.array_equal(acts, O.swarm_transform_actions(raw.reshape(-1, 2)).reshape(raw.shape).astype(np.float32))
        steps, E = acts.shape[:2]
        states = _dense(g[k + "states_idx"], g[k + "states_val"], g[k + "states_shape"])
        hist = _dense(g[k + "hist_idx"], g[k + "hist_val"], g[k + "hist_shape"])
        rep = _replay_swarm_worker(golden, acts, 4, E)
        n_hist = np.zeros(E, int)
        for t, (x, xa, rew, done, obs) in enumerate(rep):
            assert np.array_equal(done.astype(np.float32), g[k + "done"][t][:, 0])
            np.testing.assert_allclose(rew.astype(np.float32), g[k + "rew"][t][:, 0], rtol=1e-6)
            assert (g[k + "rew"][t] == g[k + "rew"][t][:, :1]).all() and (g[k + "done"][t] == g[k + "done"][t][:, :1]).all()
            # raw env state behind the slots: the float32 action row keeps its dtype through the wind add and dt * v
            assert np.array_equal(xa, g[k + "xa"][t])
            np.testing.assert_allclose(x, g[k + "x"][t], rtol=1e-12, atol=1e-14)
            n_hist = np.where(done, 1, n_hist + 1)
            for i in range(E):
                lb, ab, pos = obs[i]
                assert np.array_equal(pos, g[k + "pos"][t][i])
                local = O.swarm_local_states(O.swarm_grid_from_compact(lb, ab, 84), pos)
                assert np.array_equal(local, states[t, i])
                assert np.array_equal(O.swarm_history_window(local, n_hist[i], rnn), hist[t, i])
        assert g[k + "done"][3].all() and g[k + "done"].sum() == E * 10
