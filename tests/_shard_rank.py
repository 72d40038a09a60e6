"""Rank process of tests/test_gpu_sharded.py (started by goldsrl.distributed.spawn_local_ranks): owns E_total/world envs of one
PAAC job on device 0, exchanges the gradient with the other rank and writes its parameters after `updates` updates."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "golds-rl-gym_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_rollout(kind, E, T, env_id_offset, device=0):
    from goldsrl import _ffi
    from goldsrl import rollout as R
    if kind == "conv":
        eng = _ffi.Engine(_ffi.ENV_SWARM, E, device_id=device, seed=1692, env_id_offset=env_id_offset)
        eng.reset()
        roll = R.ConvPolicyRollout(eng, T, lr=1e-4, chunk=160)      # several chunks per step, ragged last one when E*10 % 160 != 0
    elif kind == "solow":
        eng = _ffi.Engine(_ffi.ENV_SOLOW, E, device_id=device, seed=1692, env_id_offset=env_id_offset)
        eng.reset()
        roll = R.FlatPolicyRollout(eng, T, lr=1e-3)
    else:
        eng = _ffi.Engine(_ffi.ENV_TRADE, E, device_id=device, seed=1692, env_id_offset=env_id_offset, n_assets=16, rnn_length=20)
        eng.reset()
        roll = R.FlatPolicyRollout(eng, T, lr=1e-3)
    return eng, roll


def main():
    kind, e_total, T, updates, prefer, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    from goldsrl import distributed as D
    from goldsrl import sharding
    ranks = D.Ranks().init(timeout_s=300)
    lo, hi = sharding.shard_range(e_total, ranks.rank, ranks.world)
    eng, roll = make_rollout(kind, hi - lo, T, lo)
    exchange = D.attach_gradient_exchange(roll, ranks, prefer=prefer)
    grads = None
    for _ in range(updates):
        roll.run()
        if grads is None:
            grads = roll.net.get_grads()        # host path: the summed gradient; RCCL: the all-reduced one
    eng.wait()
    np.savez(os.path.join(out, "rank%d.npz" % ranks.rank), params=roll.net.get_params(), grads=grads,
             exchange=np.array(exchange), world=ranks.world, stats=np.array([roll.last_stats[k] for k in ("loss", "global_norm")]))
    ranks.barrier()
    roll.net.close(); eng.close()
    ranks.close()


if __name__ == "__main__":
    main()
