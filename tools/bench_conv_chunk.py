"""Full PAAC update vs chunk size (samples per pass)."""
import sys, time
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, rollout
E, T = 32768, 20
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
eng.reset()
for chunk in (40960, 81920, 131040, 40960, 81920, 131040):      # 0.577/0.581, 0.570/0.572, 0.591/0.593 s per update (round 2)
    r = rollout.ConvPolicyRollout(eng, T, train=True, chunk=chunk)
    r.run(); eng.wait()
    t0 = time.perf_counter(); r.run(); r.run(); eng.wait(); dt = (time.perf_counter() - t0) / 2
    print('chunk', chunk, 'time %.3f s' % dt, 'env-steps/s %.3e' % (E * T / dt), flush=True)
    r.net.close()
