import sys, time
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, rollout
E = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = 20
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
eng.reset()
for train in (False, True):
    r = rollout.ConvPolicyRollout(eng, T, train=train)
    r.run(); eng.wait()
    r.net.profile_enable(True)
    t0 = time.perf_counter(); r.run(); eng.wait(); dt = time.perf_counter() - t0
    n, ms, fl = r.net.profile_read()
    print('train' if train else 'rollout', 'E', E, 'time %.3f s' % dt, 'env-steps/s %.3e' % (E * T / dt),
          'gemm launches', n, 'gemm ms %.1f' % ms, 'TFLOP/s %.1f' % (fl / ms / 1e9 if ms else 0), 'gemm share %.2f' % (ms / 1e3 / dt), r.last_stats)
    r.net.close()
