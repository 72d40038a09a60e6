"""HBM in use after one full-size update (keep buffers + 4 lanes of workspace)."""
import sys, subprocess
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, rollout
eng = _ffi.Engine(_ffi.ENV_SWARM, 32768, seed=1692)
eng.reset()
r = rollout.ConvPolicyRollout(eng, 20, train=True)
r.run(); eng.wait()
out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
print(out)
