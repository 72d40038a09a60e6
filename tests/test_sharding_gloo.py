"""CPU, world_size 2, gloo: the N>1 host logic -- contiguous env shards, global-env-id generator keys
(sharded == unsharded), the id broadcast bench.py uses for RCCL bootstrap, and the gradient rule
'sum over ranks / world' for a loss that is a mean over the whole batch.  Arithmetic here is the
ORACLE's (this is a CPU test of the plan, not of the kernels)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "golds-rl-gym_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from goldsrl import sharding
    from oracle import nets as NN
    from oracle import oracle as O
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        total, seed = 8, 1692
        lo, hi = sharding.shard_range(total, rank, world)
        assert sharding.env_id_offset(rank, total // world) == lo
        # 1) unique-id style broadcast (bench.py: rank 0 makes the RCCL id, the others receive it)
        uid = np.arange(128, dtype=np.uint8) if rank == 0 else np.zeros(128, np.uint8)
        t = torch.from_numpy(uid); dist.broadcast(t, src=0)
        assert np.array_equal(t.numpy(), np.arange(128, dtype=np.uint8))
        # 2) shard-local reset draws keyed by GLOBAL env id == slice of the unsharded draws
        env = np.arange(lo, hi)
        x0_local = np.stack(O.u01_pair(O.rng_block(seed, env[:, None], 0, 0, np.arange(80)[None])), axis=-1)
        x0_full = np.stack(O.u01_pair(O.rng_block(seed, np.arange(total)[:, None], 0, 0, np.arange(80)[None])), axis=-1)
        assert np.array_equal(x0_local, x0_full[lo:hi])
        # 3) gradient of a batch-mean loss: all-reduce(sum of per-rank means) / world == full-batch gradient
        rng = np.random.RandomState(0)
        n = 4
        states = np.zeros((n, 84, 84, 3))
        for i in range(n):
            states[i, rng.randint(84), rng.randint(84), 0] = 1 / 80.0
            states[i, rng.randint(84), rng.randint(84), 2] = 1.0
        act, adv, y = rng.normal(size=(n, 2)), rng.normal(size=n) * 0.01, -rng.rand(n) * 100
        p = NN.conv_init(3)
        sl = slice(rank * n // world, (rank + 1) * n // world)
        _, _, _, g_local, _ = NN.conv_loss_and_grads(p, states[sl], act[sl], adv[sl], y[sl], 0.02, 1000.0)
        flat = torch.from_numpy(NN.flatten_params(g_local))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat = flat.numpy() / world
        _, _, _, g_full, _ = NN.conv_loss_and_grads(p, states, act, adv, y, 0.02, 1000.0)
        np.testing.assert_allclose(flat, NN.flatten_params(g_full), rtol=1e-9, atol=1e-12)
        # the clip is applied AFTER the reduction, so every rank clips the same vector
        clipped, norm = NN.clip_by_global_norm(flat, 0.01)
        tn = torch.tensor([norm]); dist.all_reduce(tn, op=dist.ReduceOp.MAX)
        assert abs(float(tn[0]) - norm) < 1e-15
        # max-over-ranks timing rule of bench.py
        tt = torch.tensor([1.0 + rank], dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        assert float(tt[0]) == float(world)
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo_sharding_plan():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_range_rules():
    sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))
    from goldsrl import sharding
    assert [sharding.shard_range(262144, r, 8) for r in (0, 7)] == [(0, 32768), (229376, 262144)]
    with pytest.raises(ValueError):
        sharding.shard_range(10, 0, 3)          # np.split semantics (runners.py:18-19)
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)
    assert sharding.global_mean_from_shards(6.0, 3, lambda v: 2 * v) == 2.0
