"""How long the host needs to ENQUEUE one full update (all launches on four streams) against how long the GPU needs to run it."""
import sys, time
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, rollout
E = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
eng.reset()
r = rollout.ConvPolicyRollout(eng, 20, train=True)
r.run(); eng.wait()
for _ in range(3):
    t0 = time.perf_counter(); r.run(); t1 = time.perf_counter(); eng.wait(); t2 = time.perf_counter()
    print("enqueue %.1f ms, until done %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
