"""GPU: the device path against fixtures captured from the reference's own learner loops and Swarm worker
(tests/golden/paac_loop.npz, swarm_runner.npz; generator tests/golden/gen_golden_learner.py runs PAACLearner.train,
GridPAACLearner.train and SwarmRunner._run unmodified under a canned network).  Rows R2, R3, R4, R6, S9, quirks Q4-Q7.

Bars: Swarm agents bit-exact, locusts 1e-12, bins / positions / dense states exact; Solow float32 dynamics, returns and
advantages 1e-5 relative; episode counters and global_step exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dense(idx, val, shape):
    a = np.zeros(tuple(int(s) for s in shape))
    a[tuple(idx.T)] = val
    return a


def _global_steps(recs, E):
    return [(int(r["step_index"]) - 1) * E + int(r["env"]) + 1 for r in recs]


def test_flat_learner_loop_replayed_on_the_device(golden):
    from goldsrl import _ffi
    g = golden("paac_loop")
    E, T, U, cap = int(g["flat_E"]), int(g["flat_T"]), int(g["flat_updates"]), int(g["flat_cap"])
    gamma, scale = float(g["flat_gamma"]), float(g["flat_scale"])
    tape = g["flat_tape"]
    eng = _ffi.Engine(_ffi.ENV_SOLOW, E, solow_tape_len=len(tape), max_episode_steps=cap, rnn_length=5,
                      flags=_ffi.F_RESET_FROM_SNAPSHOT)
    eng.set_state("SOLOW_Z0", np.tile(g["flat_z0"], (E, 1)))
    eng.reset()
    eng.set_state("SOLOW_TAPE", np.tile(tape, (E, 1)))
    eng.episodes_enable(capacity=4 * E)
    keep = g["flat_shared_states"].shape[1]
    np.testing.assert_allclose(eng.read("obs")[:keep], g["flat_shared_states"][0], rtol=1e-6)
    raw = g["flat_actions"].reshape(U * T, E, 1)
    rew, done = np.zeros((U * T, E), np.float32), np.zeros((U * T, E), np.float32)
    for i in range(U * T):
        a = eng.transform_actions(raw[i].astype(np.float32))                 # SolowRunner.transform_actions_for_env
        np.testing.assert_allclose(a, g["flat_post_act"][i], rtol=3e-7)
        eng.step(g["flat_post_act"][i])                                     # teacher-forced: the reference's own action slot
        np.testing.assert_allclose(eng.read("obs"), g["flat_post_states"][i], rtol=1e-5, atol=1e-6)
        rew[i] = eng.read("reward"); done[i] = eng.read("done")
        np.testing.assert_allclose(rew[i], g["flat_post_rew"][i], rtol=1e-5, atol=2e-6)
        assert np.array_equal(done[i], g["flat_post_done"][i])
        if i + 1 < U * T:
            h = eng.read("history")[:keep]
            np.testing.assert_allclose(h, g["flat_shared_hist"][i + 1], rtol=1e-5, atol=1e-6)
            assert np.array_equal(h == 0, g["flat_shared_hist"][i + 1] == 0)
    # R2 + R4: masked, clipped n-step returns of every update from the device's own rewards / dones
    for u in range(U):
        sl = slice(u * T, (u + 1) * T)
        y, adv = eng.returns(rew[sl], g["flat_vs"].reshape(U, T, E)[u], g["flat_boot"][u], gamma, masks=1.0 - done[sl],
                             scale=scale, clip=(-2.0, 2.0))
        np.testing.assert_allclose(y, g["flat_y_batch"][u], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(y.reshape(-1), g["flat_feed_critic_target"][u], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(adv.reshape(-1), g["flat_feed_advantages"][u], rtol=1e-5, atol=2e-6)
    # R6: the records the loop handed its summary writer
    recs = eng.episodes_read()
    assert _global_steps(recs, E) == list(g["flat_rl_step"]) and [int(r["env"]) for r in recs] == list(g["flat_rl_env"])
    assert all(int(r["length"]) == cap for r in recs)
    np.testing.assert_allclose([r["total_reward"] for r in recs], g["flat_rl_reward"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose([r["total_reward"] / r["length"] for r in recs], g["flat_total_rewards_final"], rtol=1e-5, atol=1e-6)
    tot, steps = eng.episodes_running()
    np.testing.assert_allclose(tot, g["flat_running_total"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(steps, g["flat_running_steps"])
    eng.close()


def _seeded_swarm_engine(golden, E, cap):
    from goldsrl import _ffi
    r = golden("swarm_reset")
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, max_episode_steps=cap, flags=_ffi.F_RESET_FROM_SNAPSHOT)
    t = lambda a: np.tile(a[None], (E, 1, 1))
    for f, a in (("SWARM_X", r["s192_x"]), ("SWARM_XA", r["s192_xa"]), ("RESET_X", r["s192_x"]), ("RESET_XA", r["s192_xa"]),
                 ("SWARM_PNOISE", r["s192_particle_noise"][10]), ("SWARM_ANOISE", r["s192_agent_noise"][10]),
                 ("RESET_PNOISE", r["s192_particle_noise"][10]), ("RESET_ANOISE", r["s192_agent_noise"][10])):
        eng.set_state(f, t(a))
    eng.observe()
    return eng


def test_swarm_runner_slots_on_the_device(golden):
    """S9 against SwarmRunner._run itself: the float32 action row keeps its dtype through wind and dt*v (agents bit-exact),
    TimeLimit(4), reset observation on done, dense STATE rows, positions, and the HISTORY slot of the boundary."""
    from oracle import oracle as O
    g = golden("swarm_runner")
    for rnn in (1, 2):
        k = "r%d_" % rnn
        acts = g[k + "act"]
        steps, E = acts.shape[:2]
        eng = _seeded_swarm_engine(golden, E, 4)
        states = _dense(g[k + "states_idx"], g[k + "states_val"], g[k + "states_shape"])
        hist = _dense(g[k + "hist_idx"], g[k + "hist_val"], g[k + "hist_shape"])
        init = _dense(g[k + "init_states_idx"], g[k + "init_states_val"], (E, 10, 84, 84, 3))
        assert np.array_equal(eng.materialize_states(), init.astype(np.float32))
        assert np.array_equal(eng.read("positions"), g[k + "init_pos"])
        n_hist = np.zeros(E, int)
        for t in range(steps):
            a = eng.transform_actions(g[k + "raw_actions"][t].reshape(-1, 2).astype(np.float32)).reshape(E, 10, 2)
            np.testing.assert_allclose(a, acts[t], rtol=3e-7, atol=1e-9)
            eng.step(acts[t])
            assert np.array_equal(eng.get_state("SWARM_XA"), g[k + "xa"][t])
            np.testing.assert_allclose(eng.get_state("SWARM_X"), g[k + "x"][t], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(eng.read("reward"), g[k + "rew"][t][:, 0], rtol=1e-6)
            assert np.array_equal(eng.read("done").astype(np.float32), g[k + "done"][t][:, 0])
            assert np.array_equal(eng.read("positions"), g[k + "pos"][t])
            dense = eng.materialize_states()
            assert np.array_equal(dense, states[t].astype(np.float32))
            n_hist = np.where(eng.read("done") != 0, 1, n_hist + 1)
            for i in range(E):
                assert np.array_equal(O.swarm_history_window(dense[i].astype(np.float64), n_hist[i], rnn).astype(np.float32),
                                      hist[t, i].astype(np.float32))
        eng.close()


def test_grid_learner_loop_replayed_on_the_device(golden):
    """GridPAACLearner.train's slots and arrays: STATE / POSITIONS / REWARD / DONE of every step, unmasked unclipped returns
    over the Q4 reward layout, feed order, and the R6 records."""
    g = golden("paac_loop")
    E, T, U, cap = int(g["grid_E"]), int(g["grid_T"]), int(g["grid_updates"]), int(g["grid_cap"])
    B = E * 10
    gamma, scale = float(g["grid_gamma"]), float(g["grid_scale"])
    eng = _seeded_swarm_engine(golden, E, cap)
    eng.episodes_enable(capacity=4 * E)
    st = _dense(g["grid_shared_states_idx"], g["grid_shared_states_val"], g["grid_shared_states_shape"])
    assert np.array_equal(eng.materialize_states(), st[0].astype(np.float32))
    assert np.array_equal(eng.read("positions"), g["grid_shared_pos"][0])
    rew = np.zeros((U * T, E), np.float32)
    for i in range(U * T):
        eng.step(g["grid_shared_act"][i + 1])            # the action slot as the workers read it for step i
        rew[i] = eng.read("reward")
        np.testing.assert_allclose(rew[i], g["grid_shared_rew"][i + 1][:, 0], rtol=1e-6)
        assert np.array_equal(eng.read("done").astype(np.float32), g["grid_shared_done"][i + 1][:, 0])
        assert np.array_equal(eng.read("positions"), g["grid_shared_pos"][i + 1])
        assert np.array_equal(eng.materialize_states(), st[i + 1].astype(np.float32))
    for u in range(U):
        r_ref = np.zeros((T, B), np.float32)
        r_ref[:, :E] = rew[u * T:(u + 1) * T]            # quirk Q4: rewards[t, e_idx], e_idx < E (paac.py:338)
        y, adv = eng.returns(r_ref, g["grid_vs"].reshape(U, T, B)[u], g["grid_boot"][u], gamma, scale=scale)
        np.testing.assert_allclose(y, g["grid_y_batch"][u], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(adv.reshape(-1), g["grid_feed_advantages"][u], rtol=1e-5, atol=1e-8)
    recs = eng.episodes_read()
    assert _global_steps(recs, E) == list(g["grid_rl_step"]) and [int(r["env"]) for r in recs] == list(g["grid_rl_env"])
    np.testing.assert_allclose([r["total_reward"] for r in recs], g["grid_rl_reward"], rtol=2e-6)
    tot, steps = eng.episodes_running()
    np.testing.assert_allclose(tot, g["grid_running_total"], rtol=2e-6)
    assert np.array_equal(steps, g["grid_running_steps"])
    eng.close()


def test_grid_runners_slots_incl_history_against_swarm_runner(golden):
    """The drop-in GridRunners (goldsrl.agents.paac.runners) over SEEDED facade envs against SwarmRunner._run itself with
    rnn_length 2: STATE, HISTORY (the worker's window as it behaves: copies of the current state, int-truncated while short),
    POSITIONS, REWARD, DONE of every step incl. the TimeLimit reset, which replays the reference's seed-192 episode."""
    from goldsrl.agents.paac.emulator_runner import SwarmRunner
    from goldsrl.agents.paac.runners import GridRunners
    from goldsrl.agents.state_processors import SwarmStateProcessor
    from goldsrl.envs.multiagent import SwarmEnv
    g = golden("swarm_runner")
    rnn, k = 2, "r2_"
    acts = g[k + "act"]
    steps, E = acts.shape[:2]
    emulators = np.asarray([SwarmEnv(seed=192, max_episode_steps=4) for _ in range(E)])
    sp = SwarmStateProcessor(grid_size=84)
    init, idxs = [], []
    for em in emulators:                            # paac.py:245-251
        s = sp.process_state(em.reset())
        init.append(SwarmRunner.get_local_states(s, sp.positions)); idxs.append(sp.positions)
    init = np.array(init)
    assert np.array_equal(init, _dense(g[k + "init_states_idx"], g[k + "init_states_val"], (E, 10, 84, 84, 3)))
    hist0 = np.zeros((E, 10, rnn, 84, 84, 3), np.float32); hist0[:, :, 0] = init
    variables = [init, hist0, np.array(idxs), np.zeros((E, 10), np.float32), np.zeros((E, 10), np.float32), np.zeros((E, 10, 2), np.float32)]
    runners = GridRunners(emulators, 2, variables, SwarmRunner, None, 84)
    states, hists, positions, rewards, overs, actions = runners.get_shared_variables()
    ref_states = _dense(g[k + "states_idx"], g[k + "states_val"], g[k + "states_shape"])
    ref_hist = _dense(g[k + "hist_idx"], g[k + "hist_val"], g[k + "hist_shape"])
    for t in range(steps):
        for i in range(E):
            actions[i] = acts[t][i]                 # in place, as paac.py:313-314
        runners.update_environments(); runners.wait_updated()
        assert np.array_equal(states, ref_states[t])
        assert np.array_equal(hists, ref_hist[t].astype(np.float32))
        assert np.array_equal(positions, g[k + "pos"][t])
        np.testing.assert_allclose(rewards, g[k + "rew"][t], rtol=1e-6)
        assert np.array_equal(overs, g[k + "done"][t])
    assert overs.sum() == 0 and g[k + "done"][3].all()
    runners.stop()
