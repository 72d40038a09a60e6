"""Soak: the training script's learner for a few hundred updates with monitor, checkpoints and summaries; host RSS and
device memory must not grow."""
import os, sys, resource, subprocess, tempfile, time
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl.scripts import train_paac_conv as S
from goldsrl.agents.paac.emulator_runner import SwarmRunner
from goldsrl.agents.paac.paac import GridPAACLearner

def vram():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
    return int([l for l in out.splitlines() if "Used" in l][0].split(":")[-1])

tmp = tempfile.mkdtemp()
E, T, U = 2048, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 300
args = S.get_arg_parser().parse_args(["-ec", str(E), "--max_local_steps", str(T), "--max_global_steps", str(E * T * U), "--eval-every", "2",
                                      "--checkpoint-every", "25", "--checkpoint-path", os.path.join(tmp, "ck.npz"), "-df", os.path.join(tmp, "logs")])
nc, ec = S.get_network_and_environment_creator(args)
learner = GridPAACLearner(nc, ec, args, SwarmRunner, state_processor=None)
marks = []
orig = learner._log_update
def hook(stats):
    orig(stats)
    n = len(marks)
    if n % 50 == 0:
        marks.append((n, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024, vram() >> 20, stats["loss"]))
        print("update %d: maxrss %d MB, vram %d MB, loss %.4f" % marks[-1], flush=True)
    else:
        marks.append(None)
learner._log_update = hook
t0 = time.time()
learner.train()
print("done: %d updates in %.1f s, global_step %d" % (U, time.time() - t0, learner.global_step))
real = [m for m in marks if m]
assert real[-1][2] - real[1][2] < 64, "device memory grew"
assert real[-1][1] - real[1][1] < 200, "host memory grew"
