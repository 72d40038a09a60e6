"""Full PAAC update for the current GRL_NET_LANES setting (no per-kernel profiling)."""
import sys, time, os
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, rollout
E = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = 20
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
eng.reset()
for train in (False, True):
    r = rollout.ConvPolicyRollout(eng, T, train=train)
    r.run(); eng.wait()
    t0 = time.perf_counter(); r.run(); r.run(); eng.wait(); dt = (time.perf_counter() - t0) / 2
    print('lanes', os.environ.get('GRL_NET_LANES', 'default'), 'train' if train else 'rollout', 'time %.3f s' % dt, 'env-steps/s %.3e' % (E * T / dt), flush=True)
    r.net.close()
