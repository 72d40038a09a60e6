"""A/B of the two-stream (lane) execution: full PAAC update without per-kernel event profiling."""
import sys, time
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, rollout
E = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = 20
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
eng.reset()
for flags in (4, 0):
    for train in (False, True):
        r = rollout.ConvPolicyRollout(eng, T, train=train, reserved=flags)
        r.run(); eng.wait()
        t0 = time.perf_counter(); r.run(); r.run(); eng.wait(); dt = (time.perf_counter() - t0) / 2
        print('flags', flags, 'train' if train else 'rollout', 'E', E, 'time %.3f s' % dt, 'env-steps/s %.3e' % (E * T / dt), r.last_stats, flush=True)
        r.net.close()
