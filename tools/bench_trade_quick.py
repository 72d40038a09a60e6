"""BASELINE config 5 (per GPU share): TradeAR1 n=16, GRU policy, T=20."""
import sys, time
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, _ffi_flat
n, R, T = 16, 20, 20
S = 1 + 2 * n
for E in (8192, 65536):
    eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=1692, n_assets=n, rnn_length=R)
    eng.reset()
    net = _ffi_flat.FlatNet(eng, static_size=S, temporal_size=S, rnn_length=R, num_actions=n, max_samples=E * T)
    net.set_params(_ffi_flat.default_init_flat(3, static_size=S, temporal_size=S, num_actions=n))
    for train in (False, True):
        net.rollout(T)
        if train: net.train_rollout(1e-4)
        eng.wait()
        K = 3
        t0 = time.perf_counter()
        for _ in range(K):
            net.rollout(T)
            if train: net.train_rollout(1e-4)
        eng.wait()
        dt = (time.perf_counter() - t0) / K
        print('TradeAR1-16 E=%d T=%d %s: %.2f ms per update, %.3e env-steps/s' % (E, T, 'rollout+train' if train else 'rollout only', dt * 1e3, E * T / dt), flush=True)
    net.close(); eng.close()
