import sys, time, numpy as np
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi
E = 32768
for flags, name in ((0, 'exact'), (_ffi.F_SWARM_FAST_MATH, 'fast'), (_ffi.F_SWARM_FAST_MATH | _ffi.F_SWARM_NO_OBSERVE, 'fast-noobs')):
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1, flags=flags)
    eng.reset()
    n = E * 20
    p = eng.dev_alloc(n * 4)
    eng.dev_randn(p, n, 2, 0)
    eng._check(eng.lib.grl_transform_actions_device(eng.h, p, E * 10))
    for _ in range(3): eng.step_device(p)
    eng.wait()
    K = 20
    eng.timer_start()
    for _ in range(K): eng.step_device(p)
    eng.timer_stop()
    ms = eng.timer_ms() / K
    print(name, 'ms/step %.3f' % ms, 'env-steps/s %.3e' % (E / ms * 1e3), 'GB/s %.1f' % (E * 4613 / ms / 1e6), 'pairs/s %.3e' % (E * 7200 / ms * 1e3))
