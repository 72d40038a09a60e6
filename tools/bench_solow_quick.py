"""BASELINE config 2: Solow-v0, 4 096 envs, 20-step PAAC rollout + FlatPolicyVNetwork, 1 GPU (launch-bound)."""
import sys, time
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, _ffi_flat
for E in (4096, 65536):
    T = 20
    eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=1692)
    eng.reset()
    net = _ffi_flat.FlatNet(eng, max_samples=E * T)
    net.set_params(_ffi_flat.default_init_flat(3))
    for train in (False, True):
        for _ in range(2):
            net.rollout(T)
            if train: net.train_rollout(1e-4)
        eng.wait()
        K = 10
        t0 = time.perf_counter()
        for _ in range(K):
            net.rollout(T)
            if train: net.train_rollout(1e-4)
        eng.wait()
        dt = (time.perf_counter() - t0) / K
        print('Solow E=%d T=%d %s: %.3f ms per update, %.3e env-steps/s' % (E, T, 'rollout+train' if train else 'rollout only', dt * 1e3, E * T / dt))
