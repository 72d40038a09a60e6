"""The host-side mirror of the reference API (goldsrl.envs / agents.*), exercised the way the reference's
own tests exercise fed_gym (tests/env_tests.py) and the way PAACLearner drives Runners (paac.py:86-137)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_swarm_env_like_reference_env_tests():
    from goldsrl.envs import multiagent
    env = multiagent.SwarmEnv()
    env.reset()
    rng = np.random.RandomState(0)
    for _ in range(20):                              # tests/env_tests.py:10-25
        state, reward, done, _ = env.step(rng.normal(size=(env.N_AGENTS, 2)))
    assert len(state) == 2 and state[0].shape == (80, 2) and state[1].shape == (10, 2)
    assert reward < 0. and not done


def test_bounding_box_like_reference_env_tests():
    from goldsrl.agents.state_processors import SwarmStateProcessor
    from goldsrl.envs import multiagent
    sp = SwarmStateProcessor()                      # grid_size=20 default, as in tests/env_tests.py:27-44
    env = multiagent.SwarmEnv()
    env.reset()
    for _ in range(200):
        state, reward, done, _ = env.step(np.zeros((10, 2)))
    grid = sp.process_state(state)
    assert grid.shape == (20, 20, 2) and abs(grid[:, :, 1].sum() - 1) < 1e-9
    max_x, max_y = state[0].max(axis=0); min_x, min_y = state[0].min(axis=0)
    mean_x = np.mean(state[0], axis=0)[0]
    assert max_x < mean_x + 1.5 and min_x > mean_x - 1.5 and max_y < 6 and min_y >= 0
    lb, ab, pos = O.swarm_observe_compact(state[0], state[1], 20)
    assert np.array_equal(grid, O.swarm_grid_from_compact(lb, ab, 20)) and np.array_equal(sp.positions, pos)


def test_make_applies_time_limit_and_eval_env_is_reseeded():
    from goldsrl import envs
    env = envs.make("Swarm-eval-v0")
    s0 = [a.copy() for a in env.reset()]
    for i in range(128):
        s, r, d, _ = env.step(np.zeros((10, 2)))
        assert d == (i == 127)                      # TimeLimit(128)
    s1 = env.reset()
    assert np.array_equal(s0[0], s1[0])             # seed=192 env repeats its episode
    with pytest.raises(KeyError):
        envs.make("Nope-v0")


def test_trade_env_like_reference_env_tests():
    from goldsrl.envs import fed_env
    env = fed_env.TradeAR1Env(std_p=0.05)
    env.reset()
    for _ in range(100):                            # deplete_test (tests/env_tests.py:92-103)
        state, reward, done, _ = env.step(np.array([0.1, 0.1]))
    assert not done and state[0] <= env.MIN_CASH and (state[1:3] > 0).all()
    env.reset()
    env.step(np.array([0.1, 0.1]))
    state, _, _, _ = env.step(np.array([-1., -1.]))  # buysell_test
    np.testing.assert_array_almost_equal(0, state[1:3])
    env.reset()
    p = [env.step(np.array([0.0, 0.0]))[0][3:5] for _ in range(100)]     # prices_test
    np.testing.assert_array_less(np.std(p, axis=0), 0.05 * 2)
    with pytest.raises(AssertionError):
        env.step(np.array([2.0, 0.0]))


def test_solow_env_like_reference_env_tests():
    from goldsrl.envs import fed_env
    env = fed_env.SolowEnv(p=3, q=2)
    env.reset()
    for _ in range(100):                            # arima_test
        state, consumption, done, _ = env.step(0.1)
    assert not done
    static = fed_env.SolowEnv(sigma=0.0, p=1, q=0, T=10000)     # SolowSSEnv's dynamics (fed_env.py:253-265)
    static.reset()
    for _ in range(10000):
        state, _, done, _ = static.step(0.1)
    np.testing.assert_allclose(state[0], (0.1 / 0.02) ** (1 / (1 - 0.33)), rtol=1e-5)


def test_runners_drop_in_protocol_solow(golden):
    """PAACLearner's use of Runners (paac.py:86-137): build variables from reset envs, write actions in
    place, update_environments / wait_updated, read the shared arrays."""
    from goldsrl.agents.paac.emulator_runner import SolowRunner
    from goldsrl.agents.paac.environment_creator import SolowEnvironmentCreator
    from goldsrl.agents.paac.runners import Runners
    from goldsrl.agents.state_processors import SolowStateProcessor
    E, rnn = 8, 5
    creator = SolowEnvironmentCreator(1, 1)
    emulators = np.asarray([creator.create_environment() for _ in range(E)])
    sp = SolowStateProcessor()
    initial_states = [sp.process_state(e.reset()) for e in emulators]
    hist = np.zeros((E, rnn, 2), np.float32); hist[:, 0] = np.array(initial_states)
    variables = [np.array(initial_states), hist, np.zeros(E, np.float32), np.zeros(E, np.float32), np.zeros((E, 1), np.float32)]

    class Coord(object):
        def should_stop(self): return False
    runners = Runners(emulators, 4, variables, SolowRunner, Coord())
    runners.start()
    states, histories, rewards, overs, actions = runners.get_shared_variables()
    k = states[:, 0].astype(np.float64) * 100; z = states[:, 1:2].astype(np.float64)
    rng = np.random.RandomState(0)
    tape = runners.engine.get_state("SOLOW_TAPE").astype(np.float64)
    e = np.zeros((E, 1)); rho_z, rho_e = O.solow_rhos(1, 1)
    for t in range(6):
        a = SolowRunner.transform_actions_for_env(rng.normal(size=(E, 1)).astype(np.float32))
        for i in range(E):
            actions[i] = a[i]                      # in place, as paac.py:127-128
        runners.update_environments(); runners.wait_updated()
        k, z, e, obs, rew = O.solow_step(k, z, e, tape[:, 2047 - t], a[:, 0].astype(np.float64), rho_z, rho_e)
        np.testing.assert_allclose(states, O.solow_process_state(obs), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rewards, rew, rtol=1e-5, atol=5e-6)
        assert not overs.any()
        n = min(t + 1, rnn)
        np.testing.assert_allclose(histories[:, :n], np.repeat(states[:, None], n, 1), rtol=1e-6)
        assert (histories[:, n:] == 0).all()
    with pytest.raises(ValueError):
        Runners(emulators, 3, variables, SolowRunner, Coord())     # np.split: unequal division
    runners.stop()


def test_grid_runners_drop_in_protocol_swarm():
    from goldsrl.agents.paac.emulator_runner import SwarmRunner
    from goldsrl.agents.paac.environment_creator import SwarmEnvironmentCreator
    from goldsrl.agents.paac.runners import GridRunners
    from goldsrl.agents.state_processors import SwarmStateProcessor
    E, G = 4, 84
    creator = SwarmEnvironmentCreator()
    assert creator.num_actions == 2
    emulators = np.asarray([creator.create_environment() for _ in range(E)])
    sp = SwarmStateProcessor(grid_size=G)
    initial_states, idxs, raw = [], [], []
    for em in emulators:                            # paac.py:245-251
        s = em.reset(); raw.append(s)
        g = sp.process_state(s)
        initial_states.append(SwarmRunner.get_local_states(g, sp.positions)); idxs.append(sp.positions)
    variables = [np.array(initial_states), None, np.array(idxs), np.zeros((E, 10), np.float32), np.zeros((E, 10), np.float32),
                 np.zeros((E, 10, 2), np.float32)]
    runners = GridRunners(emulators, 2, variables, SwarmRunner, None, G)
    states, _, positions, rewards, overs, actions = runners.get_shared_variables()
    rng = np.random.RandomState(3)
    a = SwarmRunner.transform_actions_for_env(rng.normal(size=(E * 10, 2)).astype(np.float32)).reshape(E, 10, 2)
    for i in range(E):
        actions[i] = a[i]
    runners.update_environments(); runners.wait_updated()
    x = np.stack([r[0] for r in raw]); xa = np.stack([r[1] for r in raw])
    ox, oxa, orew, _ = O.swarm_step(x, xa, a, runners.engine.get_state("SWARM_ANOISE") * 0 + np.stack(
        [em._eng.get_state("SWARM_ANOISE")[0] for em in emulators]), np.stack([em._eng.get_state("SWARM_PNOISE")[0] for em in emulators]))
    np.testing.assert_allclose(rewards, np.repeat(orew[:, None], 10, 1), rtol=1e-6)
    for i in range(E):
        lb, ab, pos = O.swarm_observe_compact(ox[i], oxa[i], G)
        assert np.array_equal(positions[i], pos)
        np.testing.assert_array_equal(states[i], O.swarm_local_states(O.swarm_grid_from_compact(lb, ab, G), pos))      # float64 slot, exact
    assert not overs.any()


def test_grid_paac_learner_runs_updates(tmp_path):
    from goldsrl.scripts import train_paac_conv as S
    ckpt = str(tmp_path / "ck.npz")
    args = S.get_arg_parser().parse_args(["-ec", "64", "--max_local_steps", "5", "--max_global_steps", "640", "--eval-every", "1e-9",
                                          "--checkpoint-every", "1", "--checkpoint-path", ckpt, "-df", str(tmp_path / "logs")])
    nc, ec = S.get_network_and_environment_creator(args)
    from goldsrl.agents.paac.emulator_runner import SwarmRunner
    from goldsrl.agents.paac.paac import GridPAACLearner
    learner = GridPAACLearner(nc, ec, args, SwarmRunner, state_processor=None)
    assert learner.get_lr() == 1e-4
    stats = learner.train()
    assert learner.global_step == 640 and np.isfinite(stats["loss"])
    assert learner.get_lr() == 1e-4 - 640 * 1e-4 / 80000000
    assert learner.rescale_reward(-5.0) == -5.0
    # the monitor played an eval episode after every update, and the last checkpoint resumes the run
    import json
    tags = [json.loads(l)["tag"] for l in open(tmp_path / "logs" / "scalars.jsonl")]
    assert tags.count("eval/total_reward") == 2 and tags.count("eval/episode_length") == 2 and tags.count("global_norm") == 2
    # ... and the same scalars sit in a TensorBoard event file (actor_learner.py:79-83 writes them with tf.summary.FileWriter)
    import glob
    from goldsrl import utils_tfevents
    ev = utils_tfevents.read_scalars(glob.glob(str(tmp_path / "logs" / "events.out.tfevents.*"))[0])
    assert [t for t, _, _, _ in ev].count("global_norm") == 2 and {s for _, _, s, _ in ev} == {320, 640}
    args2 = S.get_arg_parser().parse_args(["-ec", "64", "--max_local_steps", "5", "--max_global_steps", "960", "--eval-every", "0",
                                           "--resume", ckpt])
    nc2, ec2 = S.get_network_and_environment_creator(args2)
    learner2 = GridPAACLearner(nc2, ec2, args2, SwarmRunner, state_processor=None)
    learner2.train()
    assert learner2.global_step == 960 and learner2.network.net.get_optimizer_state()["adam_step"] == 3
    # the checkpoint carries the action-noise draw counter: the resumed run continues the stream (2 + 1 updates x 5 steps)
    assert learner.network.net.get_action_counter() == 10 and learner2.network.net.get_action_counter() == 15


def test_flat_paac_learner_runs_updates():
    from goldsrl.agents.paac.emulator_runner import SolowRunner
    from goldsrl.agents.paac.paac import PAACLearner
    from goldsrl.agents.state_processors import SolowStateProcessor
    from goldsrl.scripts import train_paac_solow as S
    args = S.get_arg_parser().parse_args(["-ec", "128", "--max_local_steps", "20", "--max_global_steps", "7680"])
    assert args.scale == 100.0 and args.rnn_length == 5
    nc, ec = S.get_network_and_environment_creator(args)
    learner = PAACLearner(nc, ec, args, SolowRunner, SolowStateProcessor())
    stats = learner.train()
    assert learner.global_step == 7680 and np.isfinite(stats["loss"])
    assert learner.rescale_reward(-5.0) == -2 and learner.rescale_reward(0.3) == 0.3      # actor_learner.py:91-97
    out = learner.network.predict(np.array([[0.65, 0.0]], np.float32), np.zeros((1, 5, 2), np.float32))
    assert set(out) == {"mu", "sigma"}


def _conv_conf():
    return dict(name='local_learning', num_actions=2, clip_norm=40.0, clip_norm_type='global', device='/gpu:0',
                entropy_regularisation_strength=0.02, scale=1000.0, height=84, width=84, channels=3)


def test_swarm_policy_monitor_eval_episode(tmp_path):
    """one seeded eval episode on Swarm-eval-v0, swarm-eval.json written, replay reproduces the score."""
    import json, queue
    from goldsrl import envs, _ffi
    from goldsrl.agents.paac import policy_monitor as PM
    from goldsrl.agents.paac.policy_v_network import ConvSingleAgentPolicyNetwork
    from goldsrl.agents.state_processors import SwarmStateProcessor
    learner_eng = _ffi.Engine(_ffi.ENV_SWARM, 2, seed=5)
    learner_eng.reset()
    global_net = ConvSingleAgentPolicyNetwork(_conv_conf()).bind(learner_eng, seed=11)
    env = envs.make("Swarm-eval-v0")
    writer = PM.ScalarWriter(str(tmp_path / "eval"))
    mon = PM.SwarmPolicyMonitor(env, global_net, SwarmStateProcessor(grid_size=84), writer, network_conf=_conv_conf())
    mon.actions_path = str(tmp_path / "swarm-eval.json")
    total, length, rewards = mon.eval_once()
    assert length == 128 and len(rewards) == 128 and np.isfinite(total) and total < 0      # TimeLimit cap; rewards are -energy
    assert np.array_equal(mon.policy_net.get_flat_params(), global_net.get_flat_params())  # copy_params_op
    saved = json.load(open(mon.actions_path))
    assert saved['score'] == total and np.asarray(saved['actions']).shape == (128, 10, 2)
    # the eval env re-seeds on reset: replaying the saved actions reproduces the episode
    q = queue.Queue()
    for a in saved['actions']:
        q.put(np.asarray(a, np.float32))
    total2, length2, rewards2 = mon.eval_once(actions=q)
    assert length2 == 128
    np.testing.assert_allclose(rewards2, rewards, rtol=1e-12)
    lines = [json.loads(l) for l in open(tmp_path / "eval" / "scalars.jsonl")]
    assert {l["tag"] for l in lines} == {"eval/total_reward", "eval/episode_length"}


def test_swarm_eval_env_plays_the_references_seed_192_episode(golden, tmp_path):
    """SURVEY 8(f) rank 1, oracle-checked: `Swarm-eval-v0` (SwarmEnv(seed=192), fed_gym/__init__.py:28-33) resets to the
    REFERENCE's state (its MT19937 draws + 10 burn-in steps: tests/golden/swarm_reset.npz) and, with the reference's scripted
    actions replayed through SwarmPolicyMonitor.eval_once(actions=queue) (policy_monitor.py:156-176), reproduces the reference's
    128-step episode of tests/golden/swarm_traj.npz step for step (rewards, done at the TimeLimit, positions every 16 steps)."""
    import queue
    from goldsrl import envs, _ffi
    from goldsrl.agents.paac import policy_monitor as PM
    from goldsrl.agents.paac.policy_v_network import ConvSingleAgentPolicyNetwork
    from goldsrl.agents.state_processors import SwarmStateProcessor
    g, r0 = golden("swarm_traj"), golden("swarm_reset")
    env = envs.make("Swarm-eval-v0")
    for _ in range(2):                     # every reset re-seeds: the same state each time
        x, xa = env.reset()
        np.testing.assert_allclose(xa, r0["s192_xa"], rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(x, r0["s192_x"], rtol=1e-11, atol=1e-12)      # 10 chained burn-in steps
    assert np.array_equal(env._eng.get_state("SWARM_ANOISE")[0], g["agent_noise_row"])      # row 10 of the reference's tables
    assert np.array_equal(env._eng.get_state("SWARM_PNOISE")[0], g["particle_noise_row"])
    # the stream continues where the reference's global generator would: its next draws are the monitor's action noise
    rs = np.random.RandomState(192)
    rs.rand(80, 2); rs.rand(10, 2); rs.normal(size=(10, 10, 2)); rs.normal(size=(138, 10, 2)); rs.normal(size=(138, 80, 2))
    assert np.array_equal(env.np_random.normal(size=(10, 2)), rs.normal(size=(10, 2)))
    # scripted episode through the monitor
    learner_eng = _ffi.Engine(_ffi.ENV_SWARM, 2, seed=5)
    learner_eng.reset()
    global_net = ConvSingleAgentPolicyNetwork(_conv_conf()).bind(learner_eng, seed=11)
    mon = PM.SwarmPolicyMonitor(env, global_net, SwarmStateProcessor(grid_size=84), PM.ScalarWriter(str(tmp_path / "eval")),
                                network_conf=_conv_conf())
    mon.actions_path = str(tmp_path / "swarm-eval.json")
    snaps = {}
    inner_step = env.step

    def recording_step(a):
        out = inner_step(a)
        snaps[len(snaps)] = (out[0][0].copy(), out[0][1].copy())
        return out
    env.step = recording_step
    q = queue.Queue()
    for a in g["actions"][:128]:
        q.put(a)
    total, length, rewards = mon.eval_once(actions=q)
    assert length == 128 and q.empty()                       # done exactly at the TimeLimit (dones[127] in the fixture)
    assert bool(g["dones"][127]) and not g["dones"][:127].any()
    np.testing.assert_allclose(rewards, g["rewards"][:128], rtol=1e-9)
    np.testing.assert_allclose(total, np.sum(g["rewards"][:128]), rtol=1e-10)
    for i, step in enumerate(g["snap_steps"]):
        if step >= 127:
            break                                             # at 127 the fixture holds the reset state (worker rule)
        np.testing.assert_allclose(snaps[int(step)][1], g["xa_snap"][i], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(snaps[int(step)][0], g["x_snap"][i], rtol=1e-7, atol=1e-9)      # 16..112 chained chaotic steps
    # and the policy-driven episode is deterministic: action noise comes from the re-seeded stream (policy_monitor.py:132)
    env.step = inner_step
    t1, l1, rew1 = mon.eval_once()
    t2, l2, rew2 = mon.eval_once()
    assert l1 == l2 == 128 and rew1 == rew2


def test_solow_policy_monitor_and_checkpoint(tmp_path):
    """Solow eval episode through FlatPolicyVNetwork.predict, and the flat-weights checkpoint round trip (8(f) rank 2)."""
    from goldsrl import _ffi, _ffi_flat
    from goldsrl.envs.fed_env import SolowEnv
    from goldsrl.agents.paac import policy_monitor as PM
    from goldsrl.agents.paac.policy_v_network import FlatPolicyVNetwork
    from goldsrl.agents.state_processors import SolowStateProcessor
    conf = dict(name='local_learning', num_actions=1, clip_norm=40.0, clip_norm_type='global', device='/gpu:0', scale=100.0,
                static_size=2, temporal_size=2, entropy_regularisation_strength=0.02, static_hidden_size=32, rnn_hidden_size=32)
    eng = _ffi.Engine(_ffi.ENV_SOLOW, 64, seed=3, max_episode_steps=1024)
    eng.reset()
    global_net = FlatPolicyVNetwork(conf).bind(eng, max_samples=64 * 8)
    env = SolowEnv(p=1, q=1, T=16, seed=1692, max_episode_steps=16)
    mon = PM.SolowPolicyMonitor(env, global_net, SolowStateProcessor(), PM.ScalarWriter(str(tmp_path / "e")), network_conf=conf)
    np.random.seed(1)
    total, length, rewards = mon.eval_once()
    assert length == 16 and np.isfinite(total)
    # checkpoint: two updates, save, two more, restore, the same two again -> identical parameters
    net = global_net.net
    for _ in range(2):
        net.rollout(8); net.train_rollout(1e-3)
    eng.wait()
    path = str(tmp_path / "ckpt.npz")
    net.save_checkpoint(path, global_step=1234)
    state = {n: eng.get_state(n) for n in ("SOLOW_K", "SOLOW_Z", "SOLOW_E")} if hasattr(eng, "get_state") else None
    p_saved, o_saved = net.get_params(), net.get_optimizer_state()
    assert o_saved["adam_step"] == 2 and np.abs(o_saved["adam_m"]).max() > 0
    for _ in range(2):
        net.rollout(8); net.train_rollout(1e-3)
    assert not np.array_equal(net.get_params(), p_saved)
    extra = net.load_checkpoint(path)
    assert int(extra["global_step"]) == 1234
    assert np.array_equal(net.get_params(), p_saved)
    o2 = net.get_optimizer_state()
    assert o2["adam_step"] == 2 and np.array_equal(o2["adam_m"], o_saved["adam_m"]) and np.array_equal(o2["adam_v"], o_saved["adam_v"])
