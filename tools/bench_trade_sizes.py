"""TradeAR1-16 rollout (GRU policy, T=20) at several env counts: where the small-grid forward (GRL_FLAT_WLDS_GROUPS) stops paying."""
import sys, time
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, _ffi_flat
n, R, T = 16, 20, 20
S = 1 + 2 * n
for E in (8192, 16384, 24576, 32768, 49152):
    eng = _ffi.Engine(_ffi.ENV_TRADE, E, seed=1692, n_assets=n, rnn_length=R)
    eng.reset()
    net = _ffi_flat.FlatNet(eng, static_size=S, temporal_size=S, rnn_length=R, num_actions=n, max_samples=E * T)
    net.set_params(_ffi_flat.default_init_flat(3, static_size=S, temporal_size=S, num_actions=n))
    net.rollout(T); eng.wait()
    t0 = time.perf_counter()
    for _ in range(5):
        net.rollout(T)
    eng.wait()
    dt = (time.perf_counter() - t0) / 5
    print('E=%d groups=%d rollout %.2f ms %.3e env-steps/s' % (E, E // 64, dt * 1e3, E * T / dt), flush=True)
    net.close(); eng.close()
