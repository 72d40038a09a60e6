"""GPU: the sharded PAAC update THROUGH THE C ABI.  The reference's loss is a mean over the whole T*B batch
(fed_gym/agents/paac/policy_v_network.py:54,62 conv; :246-251 flat), so with the env batch split over ranks
    sum over ranks of (gradient of the rank's mean loss) / world == full-batch gradient,
the clip is applied after the reduction and Adam runs replicated.  Checked for the conv net and the flat GRU net
(a) in one process with two half-size handles (train_rollout_grads -> sum -> set_grads -> apply_grads(lr, 1/2)), and
(b) with two fresh rank processes on device 0 exchanging through the host store of goldsrl.distributed (the host path bench.py falls back
to when RCCL cannot form a communicator; RCCL itself refuses two ranks on one device), against a single-process
full-batch update.  Generator streams are keyed by global env id, so the sharded rollouts ARE the full-batch rollout."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

CASES = [("conv", 24, 3), ("solow", 192, 5), ("trade", 160, 4)]


def _full_batch(kind, E, T, updates=1):
    import _shard_rank as W
    eng, roll = W.make_rollout(kind, E, T, 0)
    grads = None
    for _ in range(updates):
        roll.run()
        if grads is None:
            grads = roll.net.get_grads()
    eng.wait()
    out = roll.net.get_params(), grads, roll.last_stats
    roll.net.close(); eng.close()
    return out


def _close(a, b, lr):
    # first Adam steps move a parameter by lr * g / (|g| + 3.2e-7): where |g| is far above 3e-7 both runs move it by the same
    # amount; elsewhere a 1e-9 difference in g shows up as a fraction of lr
    assert np.abs(a - b).max() <= 0.05 * lr, np.abs(a - b).max()


@pytest.mark.parametrize("kind,E,T", CASES)
def test_two_half_handles_sum_to_the_full_batch_update(kind, E, T):
    import _shard_rank as W
    p_full, g_full, st_full = _full_batch(kind, E, T)
    halves = [W.make_rollout(kind, E // 2, T, off) for off in (0, E // 2)]
    lr = halves[0][1].lr
    local = []
    for eng, roll in halves:
        roll.net.rollout(T, 0) if kind == "conv" else roll.net.rollout(T)
        st = roll.net.train_rollout_grads()
        local.append((roll.net.get_grads(), st))
    summed = local[0][0] + local[1][0]
    # the gradient rule itself
    scale = np.abs(g_full).max()
    assert np.abs(summed * 0.5 - g_full).max() <= 2e-5 * scale, np.abs(summed * 0.5 - g_full).max() / scale
    # losses are means over the shard: the mean of the two equals the full-batch loss
    np.testing.assert_allclose(0.5 * (local[0][1]["loss"] + local[1][1]["loss"]), st_full["loss"], rtol=2e-5, atol=1e-6)
    for eng, roll in halves:
        roll.net.set_grads(summed)
        st = roll.net.apply_grads(lr, 0.5)
        np.testing.assert_allclose(st["global_norm"], st_full["global_norm"], rtol=1e-4)      # norm of the MEAN gradient
    pa, pb = halves[0][1].net.get_params(), halves[1][1].net.get_params()
    assert np.array_equal(pa, pb)              # replicated Adam on the same summed gradient: bitwise equal replicas
    _close(pa, p_full, lr)
    for eng, roll in halves:
        roll.net.close(); eng.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,E,T", CASES)
def test_two_rank_processes_host_exchange_equals_full_batch(kind, E, T, tmp_path):
    from goldsrl import distributed as D
    updates = 2
    rc = D.spawn_local_ranks([sys.executable, os.path.join(ROOT, "tests", "_shard_rank.py"), kind, str(E), str(T), str(updates), "host",
                              str(tmp_path)], 2)
    assert rc == 0
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert str(r0["exchange"]) == "host-store-fallback" and int(r0["world"]) == 2
    assert np.array_equal(r0["params"], r1["params"])
    p_full, g_full, st_full = _full_batch(kind, E, T, updates)
    scale = np.abs(g_full).max()
    assert np.abs(r0["grads"] * 0.5 - g_full).max() <= 2e-5 * scale
    lr = 1e-4 if kind == "conv" else 1e-3
    assert np.abs(r0["params"] - p_full).max() <= 0.05 * lr * updates


@pytest.mark.timeout(900)
def test_rccl_attempt_on_one_device_falls_back_loudly_not_silently(tmp_path):
    # two ranks on ONE device: RCCL must refuse (duplicate GPU) and every rank must agree on the host path
    from goldsrl import distributed as D
    rc = D.spawn_local_ranks([sys.executable, os.path.join(ROOT, "tests", "_shard_rank.py"), "solow", "64", "3", "1", "rccl", str(tmp_path)], 2)
    assert rc == 0
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert str(r0["exchange"]) == str(r1["exchange"]) == "host-store-fallback"
    assert np.array_equal(r0["params"], r1["params"])


@pytest.mark.timeout(900)
def test_bench_gpus_2_without_a_launcher_reports_two_ranks():
    env = dict(os.environ, GRL_BENCH_FORCE_DEVICE="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "256", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["gradient_exchange"] in ("rccl", "host-store-fallback")
    assert j["config"]["envs_total"] == 512 and j["value"] > 0
    assert j["strong_scaling"]["envs_total"] == 256 and j["strong_scaling"]["envs_per_gpu"] == 128
    assert j["roofline"]["frac"] > 0


@pytest.mark.timeout(900)
def test_flat_learner_sharded_over_two_ranks(tmp_path):
    """PAACLearner (Solow, FlatPolicyVNetwork) on two ranks: same wiring as the grid learner, flat net's exchange."""
    from goldsrl import distributed as D
    import _learner_rank as L
    E, T, U = 64, 20, 2
    rc = D.spawn_local_ranks([sys.executable, os.path.join(ROOT, "tests", "_learner_rank.py"), str(E), str(T), str(U), str(tmp_path), "flat"], 2)
    assert rc == 0
    r0, r1 = np.load(tmp_path / "learner_rank0.npz"), np.load(tmp_path / "learner_rank1.npz")
    assert str(r0["exchange"]) == "host-store-fallback" and int(r0["global_step"]) == U * 2 * E * T
    assert np.array_equal(r0["params"], r1["params"])
    single = tmp_path / "single"
    single.mkdir()
    L.run(2 * E, T, U, str(single), 1, "flat")
    s = np.load(single / "learner_rank0.npz")
    both = np.concatenate([r0["log"], r1["log"]])
    both = both[np.lexsort((both[:, 1], both[:, 0]))]
    assert both.shape == s["log"].shape and len(both) >= 2 * E
    assert np.array_equal(both[:, :3], s["log"][:, :3])
    np.testing.assert_allclose(both[:, 3], s["log"][:, 3], rtol=1e-4)
    assert np.abs(r0["params"] - s["params"]).max() <= 0.05 * 1e-4 * U


@pytest.mark.timeout(900)
def test_grid_learner_sharded_over_two_ranks(tmp_path):
    """The learner itself on two ranks (GridPAACLearner.train: engine with env_id_offset = rank * E, gradient exchange, R6 with
    the rank's offset in the reference's global_step): episode log and parameters equal the single-process run on 2E envs."""
    from goldsrl import distributed as D
    import _learner_rank as L
    E, T, U = 16, 20, 2
    rc = D.spawn_local_ranks([sys.executable, os.path.join(ROOT, "tests", "_learner_rank.py"), str(E), str(T), str(U), str(tmp_path)], 2)
    assert rc == 0
    r0, r1 = np.load(tmp_path / "learner_rank0.npz"), np.load(tmp_path / "learner_rank1.npz")
    assert str(r0["exchange"]) == "host-store-fallback" and int(r0["global_step"]) == int(r1["global_step"]) == U * 2 * E * T
    assert np.array_equal(r0["params"], r1["params"])
    single = tmp_path / "single"
    single.mkdir()
    L.run(2 * E, T, U, str(single), 1)
    s = np.load(single / "learner_rank0.npz")
    assert int(s["global_step"]) == U * 2 * E * T
    # R6 across ranks: the union of the two logs is the single-process log (global_step of the summary, global env id, length, total)
    both = np.concatenate([r0["log"], r1["log"]])
    both = both[np.lexsort((both[:, 1], both[:, 0]))]
    assert both.shape == s["log"].shape and len(both) == 5 * 2 * E          # TimeLimit 8: episodes end at steps 8..40
    assert np.array_equal(both[:, :3], s["log"][:, :3])
    np.testing.assert_allclose(both[:, 3], s["log"][:, 3], rtol=1e-6)       # from the 2nd update on the policies differ by round-off
    assert np.abs(r0["params"] - s["params"]).max() <= 0.05 * 1e-4 * U


def test_flat_net_rccl_communicator_world_size_1():
    # the RCCL calls themselves (init, broadcast, all-reduce inside train_rollout) with the one rank a one-GPU box allows
    import _shard_rank as W
    eng, roll = W.make_rollout("solow", 64, 4, 0)
    net = roll.net
    net.comm_init(net.comm_unique_id(), 0, 1)
    net.comm_broadcast_params(0)
    p0 = net.get_params()
    roll.run()
    st = roll.last_stats
    assert np.isfinite(list(st.values())).all() and st["global_norm"] > 0
    assert not np.array_equal(p0, net.get_params())
    net.comm_destroy()
    net.close(); eng.close()


@pytest.mark.timeout(900)
def test_bench_under_the_elastic_launcher_two_ranks_on_one_gpu():
    """The driver's N > 1 launch line, rehearsed with two ranks on device 0: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`.  The launcher owns MASTER_PORT; the ranks
    (which never import it) meet over goldsrl.distributed's own store, try RCCL (refused: two ranks on one device), fall back
    TOGETHER to the host exchange, and rank 0 prints one line with the communicator facts."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, GRL_BENCH_FORCE_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GRL_RDZV_KEY", "GRL_STORE_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "256", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-strong"], env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["envs_total"] == 512 and j["value"] > 0
    assert j["gradient_exchange"] in ("rccl", "host-store-fallback")
    assert j["comm"]["params_equal_across_ranks"] is True
    if j["gradient_exchange"] == "rccl":
        assert j["rccl_ranks"] == 2 and len(j["allreduce_ms"]) == 2 and all(m > 0 for m in j["allreduce_ms"])
    else:
        assert j["rccl_ranks"] == 0
    assert "ms_per_step_spread" in j and j["ms_per_step_spread"]["n"] == 2
    # what a rank's host thread costs per update, and where the rank was pinned (goldsrl/affinity.py) -- per rank in the line
    assert j["host_enqueue_ms_per_update"] > 0 and len(j["host"]["per_rank"]) == 2
    assert all(h["host_enqueue_ms_per_update"] > 0 and h["train_wait_ms"] >= 0 for h in j["host"]["per_rank"])
    assert len(j["config"]["cpu_affinity"]) == 2 and all("pinned" in a for a in j["config"]["cpu_affinity"])
