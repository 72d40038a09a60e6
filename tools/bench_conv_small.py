"""Full PAAC update at a strong-scaling shard size (E envs on one GPU) vs chunk size: with one chunk per step the four lanes cannot
overlap anything in the rollout."""
import sys, time
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, rollout
T = 20
for E in (4096, 8192, 16384):
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
    eng.reset()
    for chunk in (40960, 20480, 10240, 5120):
        if chunk > E * 10:
            continue
        r = rollout.ConvPolicyRollout(eng, T, train=True, chunk=chunk)
        r.run(); eng.wait()
        t0 = time.perf_counter(); r.run(); r.run(); r.run(); eng.wait(); dt = (time.perf_counter() - t0) / 3
        print('E', E, 'chunk', chunk, 'time %.1f ms' % (dt * 1e3), 'env-steps/s %.3e' % (E * T / dt), flush=True)
        r.net.close()
    eng.close()
