"""GPU: R6 -- the learner's per-env bookkeeping (reference fed_gym/agents/paac/paac.py:142-157 flat, :331-349 grid) kept on the
device (grl_episodes_*), against the restatement oracle/oracle.py:episode_bookkeeping fed with the SAME rollout's rewards and
dones.  TimeLimit is shortened to 8 so several episodes end inside a 20-step rollout; covers the conv rollout with both reward
layouts (quirk Q4 must not change the bookkeeping: the reference reads column 0 of the env's row), the flat rollout inside its
hipGraph, the plain step path, carry-over of the running sums between rollouts, and the learner's `rl/reward` points."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _check(recs, oracle_recs, E, steps_before):
    assert len(recs) == len(oracle_recs) and len(recs) > 0
    for r, (gstep, env, length, total) in zip(recs, oracle_recs):
        assert (int(r["step_index"]) - 1) * E + int(r["env"]) + 1 == gstep      # the reference's global_step at the summary
        assert int(r["env"]) == env and int(r["length"]) == length
        assert r["total_reward"] == total                                      # float64 sum of the same float32 rewards, same order


@pytest.mark.parametrize("layout", [0, 1])
def test_conv_rollout_bookkeeping_matches_oracle(layout):
    from goldsrl import _ffi, _ffi_net
    E, T = 70, 20          # two waves of envs, the second partial
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=11, max_episode_steps=8)
    eng.reset()
    eng.episodes_enable()
    net = _ffi_net.ConvNet(eng, max_chunk_samples=E * 10)
    net.set_params(_ffi_net.glorot_uniform_flat(3))
    total = steps = None
    gs = 0
    for _ in range(2):       # running sums and step counters carry over from one rollout to the next
        net.rollout(T, layout)
        rew = net.read_rollout("rewards", (T, E * 10))
        dones = net.read_rollout("dones", (T, E), np.uint8)
        r_env = rew[:, :E] if layout == 1 else rew[:, ::10]      # Q4: rewards[t, e_idx] = r_e for e_idx < E; broadcast otherwise
        if layout == 1:
            assert (rew[:, E:] == 0).all()
        assert dones.sum() == E * (T // 8 if _ == 0 else (2 * T) // 8 - T // 8)
        orecs, total, steps, gs = O.episode_bookkeeping(r_env, dones, total, steps, gs)
        _check(eng.episodes_read(), orecs, E, 0)
        rt, rl = eng.episodes_running()
        assert np.array_equal(rt, total) and np.array_equal(rl, steps)
    assert len(eng.episodes_read()) == 0        # the read emptied the list
    net.close(); eng.close()


def test_flat_rollout_graph_bookkeeping_matches_oracle():
    from goldsrl import _ffi
    from goldsrl import rollout as R
    E, T = 200, 20
    eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=5, max_episode_steps=8)
    eng.reset()
    roll = R.FlatPolicyRollout(eng, T, train=False)
    roll.run(); eng.wait()             # the rollout graph is captured BEFORE accounting is enabled: it must be re-captured
    eng.episodes_enable(capacity=T * E)
    total = steps = None
    gs = 0
    for _ in range(3):
        roll.run()
        rew = roll.net.read_rollout("rewards", (T, E))
        dones = (1.0 - roll.net.read_rollout("masks", (T, E))).astype(np.uint8)
        orecs, total, steps, gs = O.episode_bookkeeping(rew, dones, total, steps, gs)
        _check(eng.episodes_read(), orecs, E, 0)
    roll.net.close(); eng.close()


def test_step_path_bookkeeping_and_capacity_overflow_is_reported():
    from goldsrl import _ffi
    E = 96
    eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=2, n_assets=2, max_episode_steps=5)
    eng.reset()
    eng.episodes_enable(capacity=E)
    rng = np.random.RandomState(0)
    rews, dones = [], []
    for t in range(7):
        eng.step(rng.uniform(-1, 1, size=(E, 2)).astype(np.float32))
        rews.append(eng.read("reward")); dones.append(eng.read("done"))
    orecs, total, steps, gs = O.episode_bookkeeping(np.array(rews), np.array(dones))
    _check(eng.episodes_read(), orecs, E, 0)
    for t in range(10):      # two TimeLimit boundaries = 2E finished episodes > capacity E
        eng.step(np.zeros((E, 2), np.float32))
    with pytest.raises(_ffi.GrlError, match="dropped"):
        eng.episodes_read()
    eng.close()


def test_grid_learner_emits_rl_reward_at_the_reference_global_step(tmp_path):
    import glob
    from goldsrl import utils_tfevents
    from goldsrl.agents.paac.emulator_runner import SwarmRunner
    from goldsrl.agents.paac.paac import GridPAACLearner
    from goldsrl.scripts import train_paac_conv as S
    E, T = 32, 20
    args = S.get_arg_parser().parse_args(["-ec", str(E), "--max_local_steps", str(T), "--max_global_steps", str(2 * E * T), "--eval-every", "0",
                                          "-df", str(tmp_path / "logs")])
    args.max_episode_steps = 8
    args.reward_layout = "reference"
    nc, ec = S.get_network_and_environment_creator(args)
    learner = GridPAACLearner(nc, ec, args, SwarmRunner, state_processor=None)
    learner.train()
    assert learner.global_step == 2 * E * T
    # every env finishes at steps 8, 16 (first rollout) and 24, 32, 40 (second): 5 episodes per env, length 8 each
    log = learner.episode_log
    assert len(log) == 5 * E and all(l == 8 for _, _, l, _ in log)
    expect_steps = [(k * 8 - 1) * E + e + 1 for k in range(1, 6) for e in range(E)]
    assert [g for g, _, _, _ in log] == expect_steps
    np.testing.assert_allclose(learner.total_rewards[-10:], [tot / 8 for _, _, _, tot in log[-10:]])
    # the last rollout's rewards reproduce the totals of the episodes that lie entirely inside it (steps 24..40 -> t = 4..19)
    net = learner.network.net
    rew = net.read_rollout("rewards", (T, E * 10))[:, :E]
    for gstep, env, _, tot in log[-E:]:            # episodes ending at t = 19 cover t = 12..19
        acc = 0.0
        for t in range(12, 20):          # sequential float64 sum of the float32 rewards (paac.py:334)
            acc += float(rew[t, env])
        assert tot == acc
    ev = utils_tfevents.read_scalars(glob.glob(str(tmp_path / "logs" / "events.out.tfevents.*"))[0])
    pts = [(s, v) for t, v, s, _ in ev if t == "rl/reward"]
    assert [s for s, _ in pts] == expect_steps
    np.testing.assert_allclose([v for _, v in pts], [tot for _, _, _, tot in log], rtol=1e-6)
    learner.cleanup()
