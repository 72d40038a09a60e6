"""Probe: GEMM TFLOP/s of the forward pass vs batch size (cache residency) via the per-kernel profiler."""
import sys
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, _ffi_net
for E in (256, 1024, 4096):
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1)
    eng.reset()
    net = _ffi_net.ConvNet(eng, max_chunk_samples=min(40960, E * 10))
    net.set_params(_ffi_net.glorot_uniform_flat(3))
    for _ in range(3): net.predict()
    net.profile_enable(True)
    for _ in range(10): net.predict()
    n, ms, fl = net.profile_read()
    print('E', E, 'samples', E * 10, 'fwd GEMM launches', n, 'ms/forward %.3f' % (ms / 10), 'TFLOP/s %.1f' % (fl / ms / 1e9))
    net.close(); eng.close()
