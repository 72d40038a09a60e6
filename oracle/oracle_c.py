"""ctypes access to the C oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        _lib.oracle_swarm_step.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def swarm_step(x, xa, action_f32, an, pn, grid=84, threads=1, observe=True):
    """Batched SwarmEnv._step (+ process_state): returns x', xa', reward, lbins, abins, positions, threads used."""
    x = np.array(x, dtype=np.float64, order="C"); xa = np.array(xa, dtype=np.float64, order="C")
    E = x.shape[0]
    act = np.ascontiguousarray(action_f32, dtype=np.float32)
    an = np.ascontiguousarray(an, dtype=np.float64); pn = np.ascontiguousarray(pn, dtype=np.float64)
    rew = np.empty(E)
    lb = np.empty((E, 80, 2), np.uint8); ab = np.empty((E, 10, 2), np.uint8); pos = np.empty((E, 10, 2), np.uint8)
    used = lib().oracle_swarm_step(C.c_int(E), _p(x), _p(xa), _p(act), _p(an), _p(pn), _p(rew), C.c_int(grid),
                                   _p(lb) if observe else None, _p(ab), _p(pos), C.c_int(threads))
    return x, xa, rew, lb, ab, pos, used


def returns(r, v, boot, gamma, mask=None, scale=1.0):
    r = np.ascontiguousarray(r, np.float32); v = np.ascontiguousarray(v, np.float32); boot = np.ascontiguousarray(boot, np.float32)
    T, B = r.shape
    m = None if mask is None else np.ascontiguousarray(mask, np.float32)
    y = np.empty((T, B)); adv = np.empty((T, B))
    lib().oracle_returns(C.c_int(T), C.c_int(B), _p(r), _p(v), None if m is None else _p(m), _p(boot),
                         C.c_double(gamma), C.c_double(scale), _p(y), _p(adv))
    return y, adv
