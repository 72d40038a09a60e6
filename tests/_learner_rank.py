"""Rank process of tests/test_gpu_sharded.py::test_{grid,flat}_learner_sharded_over_two_ranks: the learner's train() on this rank's half of
the env batch (RANK / WORLD_SIZE from goldsrl.distributed.spawn_local_ranks), writes its episode log and parameters."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "golds-rl-gym_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(envs_per_rank, T, updates, out, world, kind="grid"):
    if kind == "grid":
        from goldsrl.agents.paac.emulator_runner import SwarmRunner
        from goldsrl.agents.paac.paac import GridPAACLearner
        from goldsrl.scripts import train_paac_conv as S
        args = S.get_arg_parser().parse_args(["-ec", str(envs_per_rank), "--max_local_steps", str(T), "--max_global_steps",
                                              str(updates * envs_per_rank * world * T), "--eval-every", "0", "-df", os.path.join(out, "logs")])
        args.max_episode_steps = 8
        args.device = "/gpu:0"
        nc, ec = S.get_network_and_environment_creator(args)
        learner = GridPAACLearner(nc, ec, args, SwarmRunner, state_processor=None)
    else:      # PAACLearner on Solow with FlatPolicyVNetwork (train_paac_solow.py)
        from goldsrl.agents.paac.emulator_runner import SolowRunner
        from goldsrl.agents.paac.paac import PAACLearner
        from goldsrl.agents.state_processors import SolowStateProcessor
        from goldsrl.scripts import train_paac_solow as S
        args = S.get_arg_parser().parse_args(["-ec", str(envs_per_rank), "--max_local_steps", str(T), "--max_global_steps",
                                              str(updates * envs_per_rank * world * T)])
        args.max_episode_steps = 8
        args.device = "/gpu:0"
        nc, ec = S.get_network_and_environment_creator(args)
        learner = PAACLearner(nc, ec, args, SolowRunner, SolowStateProcessor())
    if world > 1:      # both ranks of the test share device 0
        from goldsrl import distributed as D
        learner.ranks = D.Ranks().init(timeout_s=300)
        learner.ranks.local_rank = 0
    learner.train()
    rank = learner.ranks.rank if learner.ranks is not None else 0
    np.savez(os.path.join(out, "learner_rank%d.npz" % rank), params=learner.network.net.get_params(),
             log=np.array(learner.episode_log, np.float64).reshape(-1, 4), global_step=learner.global_step,
             exchange=np.array(learner.gradient_exchange))
    if learner.ranks is not None:
        learner.ranks.barrier()
    learner.cleanup()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(os.environ.get("WORLD_SIZE", "1")),
        sys.argv[5] if len(sys.argv) > 5 else "grid")
