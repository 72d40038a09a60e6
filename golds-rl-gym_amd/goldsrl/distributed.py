"""One process per GPU: rendezvous and the per-rollout gradient exchange for a sharded env batch.

The reference is single-device (fed_gym/agents/paac/actor_learner.py:70-75); its only parallelism is
`np.split(emulators, workers)` over OS processes (paac/runners.py:18-19).  Here each rank owns a contiguous
block of env ids (goldsrl/sharding.py) and a full replica of the (small) policy; the ONE collective is the sum
of the flat gradient once per rollout:

  * RCCL over xGMI (`grl_net_comm_*` / `grl_fnet_comm_*`): the all-reduce runs inside `train_rollout` on the
    handle's stream; or, when any rank cannot form the communicator (e.g. two ranks rehearsing on one GPU),
  * the host path: `train_rollout_grads -> all_reduce(sum) through torch.distributed/gloo -> set_grads ->
    apply_grads(lr, 1/world)`.

Either way: sum, scale by 1/world (the loss is a mean over the WHOLE batch, policy_v_network.py:54,62,246-251),
clip_by_global_norm after the reduction, Adam replicated, parameters broadcast from rank 0 once.
torch.distributed (gloo) is plumbing only: rendezvous, the 128-byte RCCL id, barriers, max-over-ranks timing.
"""
import os
import sys

import numpy as np


class Ranks(object):
    """rank / world / local device from the launcher's environment (torch.distributed.run or bench.py's own spawn)."""

    def __init__(self, env=None):
        env = os.environ if env is None else env
        self.rank = int(env.get("RANK", "0"))
        self.world = int(env.get("WORLD_SIZE", "1"))
        self.local_rank = int(env.get("LOCAL_RANK", str(self.rank)))
        self.dist = None

    def init(self, timeout_s=600):
        """gloo process group over 127.0.0.1 (nothing on the data path uses it)."""
        if self.world > 1 and self.dist is None:
            import datetime
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world,
                                    timeout=datetime.timedelta(seconds=timeout_s))
            self.dist = dist
        return self

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t[0])

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def attach_gradient_exchange(roll, ranks, prefer="rccl"):
    """Make `roll` (goldsrl.rollout.ConvPolicyRollout / FlatPolicyRollout) exchange gradients over `ranks` and start every
    rank from rank 0's parameters.  Returns the name of the exchange that will run: "none" | "rccl" | "gloo-host-fallback"."""
    if ranks.world <= 1:
        return "none"
    import torch
    dist, net, rank, world = ranks.dist, roll.net, ranks.rank, ranks.world
    ok = 0
    if prefer == "rccl":
        ok = 1
        try:
            nbytes = net.comm_unique_id().size
            uid = net.comm_unique_id() if rank == 0 else np.zeros(nbytes, np.uint8)
            t = torch.from_numpy(uid)
            dist.broadcast(t, src=0)
            net.comm_init(t.numpy(), rank, world)
        except Exception as e:       # noqa: BLE001 -- any failure means "no device communicator on this rank"
            sys.stderr.write("rank %d: RCCL communicator unavailable (%s)\n" % (rank, e))
            ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag[0]) == 1:
        net.comm_broadcast_params(0)
        return "rccl"
    if ok:
        net.comm_destroy()
    pt = torch.from_numpy(net.get_params())
    dist.broadcast(pt, src=0)
    net.set_params(pt.numpy())

    def host_allreduce(g):
        tg = torch.from_numpy(np.ascontiguousarray(g))
        dist.all_reduce(tg, op=dist.ReduceOp.SUM)
        return tg.numpy(), world
    roll.host_allreduce = host_allreduce
    return "gloo-host-fallback"


def spawn_local_ranks(argv, n, extra_env=None, port=None, poll_s=0.2):
    """Start `n` fresh rank processes of `argv` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) from a parent that has NOT
    touched the GPU, wait for them, and stop the others as soon as one fails (a rank that dies would otherwise leave its
    peers waiting in the rendezvous).  Children inherit stdout/stderr.  Returns the first non-zero exit code, or 0."""
    import socket
    import subprocess
    import time
    if port is None:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(poll_s)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:          # exact PIDs we started, never a pattern
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc
