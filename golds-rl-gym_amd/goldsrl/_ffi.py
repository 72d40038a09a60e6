"""ctypes binding of libgoldsrl.so (C ABI: include/goldsrl.h).  No PyTorch, no CPU fallback:
if the shared library is missing or no MI355X is usable, construction raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libgoldsrl.so")

ENV_SWARM, ENV_SOLOW, ENV_TRADE, ENV_TICKER = 0, 1, 2, 3
F_RESEED_EACH_RESET, F_RESET_FROM_SNAPSHOT, F_INJECT_NOISE, F_SWARM_FAST_MATH, F_SWARM_NO_OBSERVE, F_SOLOW_SS_RESET = 1, 2, 4, 8, 16, 32

OK, E_INVALID, E_NO_DEVICE, E_HIP, E_SIZE, E_ACTION_RANGE, E_STATE, E_COMM, E_RANGE = 0, -1, -2, -3, -4, -5, -6, -7, -8

# enum grl_field
FLD = dict(
    SWARM_X=0, SWARM_XA=1, SWARM_PNOISE=2, SWARM_ANOISE=3, RESET_X=4, RESET_XA=5, RESET_PNOISE=6, RESET_ANOISE=7,
    ELAPSED=8, EPISODE=9, SOLOW_K=16, SOLOW_Z=17, SOLOW_E=18, SOLOW_TAPE=19, SOLOW_TAPE_POS=20, SOLOW_Z0=21, NHIST=22,
    TRADE_CASH=32, TRADE_ASSETS=33, TRADE_QUANTITY=34, TRADE_PRICES=35, TRADE_NORMALS=36,
    TICKER_CASH=48, TICKER_ASSETS=49, TICKER_QUANTITY=50, TICKER_IDX=51, TICKER_START=52, TICKER_START0=53,
)


class GrlConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("env_kind", C.c_int32), ("num_envs", C.c_int32), ("device_id", C.c_int32),
        ("max_episode_steps", C.c_int32), ("grid_size", C.c_int32), ("n_assets", C.c_int32), ("solow_p", C.c_int32),
        ("solow_q", C.c_int32), ("solow_tape_len", C.c_int32), ("rnn_length", C.c_int32), ("flags", C.c_uint32),
        ("seed", C.c_uint64), ("env_id_offset", C.c_int64), ("solow_sigma", C.c_double), ("solow_delta", C.c_double),
        ("trade_std_p", C.c_double), ("trade_starting_balance", C.c_double),
    ]


class GrlOutPtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "reward", "reward_f64", "done", "elapsed", "locust_bins", "agent_bins", "positions", "obs_raw", "obs",
        "history", "done_list", "done_count")]


class GrlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libgoldsrl error %d: %s" % (code, msg))
        self.code = code


_lib = None

# name -> (restype, argtypes): every symbol include/goldsrl.h declares
_P, _I, _F, _SZ = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
SIGNATURES = {
    "grl_abi_version": (C.c_int, []),
    "grl_config_default": (C.c_int, [_I, C.POINTER(GrlConfig)]),
    "grl_create": (C.c_int, [C.POINTER(GrlConfig), C.POINTER(_P)]),
    "grl_destroy": (C.c_int, [_P]),
    "grl_last_error": (C.c_char_p, [_P]),
    "grl_reset": (C.c_int, [_P, _P, _I]),
    "grl_swarm_reset_injected": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "grl_ticker_set_table": (C.c_int, [_P, _P, C.c_int32]),
    "grl_set_state": (C.c_int, [_P, _I, _P, _SZ]),
    "grl_get_state": (C.c_int, [_P, _I, _P, _SZ]),
    "grl_step_async": (C.c_int, [_P, _P]),
    "grl_step_device": (C.c_int, [_P, _P]),
    "grl_wait": (C.c_int, [_P]),
    "grl_outputs": (C.c_int, [_P, C.POINTER(GrlOutPtrs)]),
    "grl_read_output": (C.c_int, [_P, C.c_char_p, _P, _SZ]),
    "grl_observe": (C.c_int, [_P]),
    "grl_swarm_materialize_states": (C.c_int, [_P, _I, _I, _P, _SZ]),
    "grl_transform_actions_device": (C.c_int, [_P, _P, _I]),
    "grl_transform_actions_host": (C.c_int, [_P, _P, _P, _I]),
    "grl_returns": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _P, _P]),
    "grl_returns_device": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _P, _P]),
    "grl_dev_alloc": (C.c_int, [_P, _SZ, C.POINTER(_P)]),
    "grl_dev_free": (C.c_int, [_P, _P]),
    "grl_dev_upload": (C.c_int, [_P, _P, _P, _SZ]),
    "grl_dev_download": (C.c_int, [_P, _P, _P, _SZ]),
    "grl_dev_copy": (C.c_int, [_P, _P, _P, _SZ]),
    "grl_dev_randn": (C.c_int, [_P, _P, _SZ, C.c_uint32, C.c_uint64]),
    "grl_profile_enable": (C.c_int, [_P, _I]),
    "grl_profile_read": (C.c_int, [_P, C.POINTER(_I), C.POINTER(C.c_float)]),
    "grl_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "grl_timer_start": (C.c_int, [_P]),
    "grl_timer_stop": (C.c_int, [_P]),
    "grl_timer_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "grl_swarm_step_f64": (C.c_int, [_P, _P]),
    "grl_swarm_step_opts": (C.c_int, [_P, _P, _I, _I]),
    "grl_device_pci_address": (C.c_int, [_I, C.c_char_p, _SZ]),
    "grl_episodes_enable": (C.c_int, [_P, _I]),
    "grl_episodes_read": (C.c_int, [_P, _P, _I, C.POINTER(_I), C.POINTER(_I)]),
    "grl_episodes_running": (C.c_int, [_P, _P, _P]),
}

# grl_episode_record (include/goldsrl.h)
EPISODE_DTYPE = np.dtype([("step_index", np.int64), ("env", np.int32), ("length", np.int32), ("total_reward", np.float64)])


def load_library(path=None, extra_signatures=None):
    """dlopen libgoldsrl.so and attach prototypes.  Raises OSError if it has not been built
    (python -c 'import __graft_entry__ as g; g.build()' or make -C golds-rl-gym_amd)."""
    global _lib
    if _lib is None or path is not None:
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise OSError("libgoldsrl.so not found at %s -- build it first (make -C golds-rl-gym_amd); "
                          "there is no CPU fallback" % p)
        lib = C.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    if extra_signatures:
        for name, (res, args) in extra_signatures.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


_FIELD_DTYPE = {}
for _k, _v in FLD.items():
    if _k.startswith("SWARM_") or _k.startswith("RESET_") or _k in ("TICKER_CASH", "TICKER_ASSETS", "TICKER_QUANTITY",
                                                                     "TRADE_CASH", "TRADE_ASSETS", "TRADE_QUANTITY", "TRADE_PRICES"):
        _FIELD_DTYPE[_v] = np.float64
    elif _k in ("ELAPSED", "EPISODE", "SOLOW_TAPE_POS", "NHIST", "TICKER_IDX", "TICKER_START", "TICKER_START0"):
        _FIELD_DTYPE[_v] = np.int32
    else:
        _FIELD_DTYPE[_v] = np.float32


def device_pci_address(device_id):
    """'dddd:bb:dd.f' of HIP device `device_id`, or None (no such device / no GPU)."""
    buf = C.create_string_buffer(32)
    if load_library().grl_device_pci_address(int(device_id), buf, 32) != OK:
        return None
    return buf.value.decode() or None


class Engine(object):
    """One batch of environments on one GPU: thin object wrapper over a grl_handle."""

    def __init__(self, env_kind, num_envs, device_id=0, **kw):
        self.lib = load_library()
        cfg = GrlConfig()
        rc = self.lib.grl_config_default(env_kind, C.byref(cfg))
        if rc != OK:
            raise GrlError(rc, "grl_config_default")
        cfg.num_envs, cfg.device_id = int(num_envs), int(device_id)
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown grl_config field %r" % k)
            setattr(cfg, k, v)
        self.cfg = cfg
        self.kind, self.E = env_kind, int(num_envs)
        h = C.c_void_p()
        rc = self.lib.grl_create(C.byref(cfg), C.byref(h))
        if rc != OK:
            raise GrlError(rc, self.lib.grl_last_error(None).decode())
        self.h = h
        if env_kind == ENV_SWARM:
            self.action_shape, self.obs_dim = (self.E, 10, 2), None
        elif env_kind == ENV_SOLOW:
            self.action_shape, self.obs_dim = (self.E, 1), 2
        elif env_kind == ENV_TICKER:
            self.action_shape, self.obs_dim = (self.E, 4), 7       # [choice0, choice1, fraction0, fraction1]
        else:
            self.action_shape, self.obs_dim = (self.E, cfg.n_assets), 1 + 2 * cfg.n_assets

    # -- plumbing
    def _check(self, rc):
        if rc != OK:
            raise GrlError(rc, self.lib.grl_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.grl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def field_shape(self, field):
        E, c = self.E, self.cfg
        f = FLD[field] if isinstance(field, str) else field
        inner = {0: (80, 2), 1: (10, 2), 2: (80, 2), 3: (10, 2), 4: (80, 2), 5: (10, 2), 6: (80, 2), 7: (10, 2),
                 8: (), 9: (), 16: (), 17: (c.solow_p,), 18: (max(c.solow_q, 1),), 19: (c.solow_tape_len,), 20: (),
                 21: (c.solow_p,), 22: (), 32: (), 33: (), 34: (c.n_assets,), 35: (c.n_assets,), 36: (c.n_assets,),
                 48: (), 49: (), 50: (2,), 51: (), 52: (), 53: ()}[f]
        return (E,) + inner

    # -- state
    def reset(self, idx=None):
        if idx is None:
            self._check(self.lib.grl_reset(self.h, None, 0))
        else:
            a = np.ascontiguousarray(idx, dtype=np.int32)
            self._check(self.lib.grl_reset(self.h, _ptr(a), a.size))

    def swarm_step_f64(self, actions):
        """SwarmEnv.step with float64 actions (the eval monitor's call); synchronous like step()."""
        a = np.ascontiguousarray(actions, dtype=np.float64)
        if a.shape != (self.E, 10, 2):
            raise ValueError("swarm_step_f64: expected shape %s, got %s" % ((self.E, 10, 2), a.shape))
        self._check(self.lib.grl_swarm_step_f64(self.h, _ptr(a)))
        self.wait()

    def swarm_step_opts(self, actions, add_wind=True):
        """SwarmEnv._step(v_action, add_wind) (multiagent.py:30-36): float32 actions keep the worker's float32 arithmetic, anything
        else is stepped as float64; synchronous like step()."""
        a = np.asarray(actions)
        f64 = a.dtype != np.float32
        a = np.ascontiguousarray(a, dtype=np.float64 if f64 else np.float32)
        if a.shape != (self.E, 10, 2):
            raise ValueError("swarm_step_opts: expected shape %s, got %s" % ((self.E, 10, 2), a.shape))
        self._check(self.lib.grl_swarm_step_opts(self.h, _ptr(a), 1 if f64 else 0, 1 if add_wind else 0))
        self.wait()

    def swarm_reset_injected(self, x0, xa0, random_actions, agent_noise, particle_noise):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x0, xa0, random_actions, agent_noise, particle_noise)]
        E = self.E
        want = [(E, 80, 2), (E, 10, 2), (E, 10, 10, 2), (E, 11, 10, 2), (E, 11, 80, 2)]
        for a, w in zip(arrs, want):
            if a.shape != w:
                raise ValueError("swarm_reset_injected: expected shape %s, got %s" % (w, a.shape))
        self._check(self.lib.grl_swarm_reset_injected(self.h, *[_ptr(a) for a in arrs]))

    def ticker_set_table(self, matrix):
        """Upload the OpenCloseSampler data matrix (rows, 4) every env samples its 1024-row windows from."""
        m = np.ascontiguousarray(matrix, dtype=np.float64)
        if m.ndim != 2 or m.shape[1] != 4:
            raise ValueError("ticker_set_table: expected a (rows, 4) matrix, got %s" % (m.shape,))
        self._check(self.lib.grl_ticker_set_table(self.h, _ptr(m), m.shape[0]))

    def set_state(self, field, value):
        f = FLD[field] if isinstance(field, str) else field
        a = np.ascontiguousarray(value, dtype=_FIELD_DTYPE[f])
        if a.shape != self.field_shape(f):
            raise ValueError("set_state(%s): expected shape %s, got %s" % (field, self.field_shape(f), a.shape))
        self._check(self.lib.grl_set_state(self.h, f, _ptr(a), a.nbytes))

    def get_state(self, field):
        f = FLD[field] if isinstance(field, str) else field
        a = np.empty(self.field_shape(f), dtype=_FIELD_DTYPE[f])
        self._check(self.lib.grl_get_state(self.h, f, _ptr(a), a.nbytes))
        return a

    # -- step protocol (Runners.update_environments / wait_updated)
    def step_async(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.size != int(np.prod(self.action_shape)):
            raise ValueError("actions: expected %s values, got shape %s" % (self.action_shape, a.shape))
        self._actions_keepalive = a
        self._check(self.lib.grl_step_async(self.h, _ptr(a)))

    def step_device(self, dev_ptr):
        self._check(self.lib.grl_step_device(self.h, C.c_void_p(dev_ptr)))

    def wait(self):
        self._check(self.lib.grl_wait(self.h))

    def step(self, actions):
        self.step_async(actions)
        self.wait()

    def observe(self):
        self._check(self.lib.grl_observe(self.h))

    _OUT = {"reward": (np.float32, ()), "reward_f64": (np.float64, ()), "done": (np.uint8, ()), "elapsed": (np.int32, ()),
            "locust_bins": (np.uint8, (80, 2)), "agent_bins": (np.uint8, (10, 2)), "positions": (np.uint8, (10, 2))}

    def read(self, name):
        if name in self._OUT:
            dt, inner = self._OUT[name]
            a = np.empty((self.E,) + inner, dtype=dt)
        elif name in ("obs", "obs_raw"):
            a = np.empty((self.E, self.obs_dim), dtype=np.float32)
        elif name == "history":
            a = np.empty((self.E, self.cfg.rnn_length, 2), dtype=np.float32)
        elif name == "done_count":
            a = np.empty((1,), dtype=np.int32)
        elif name == "done_list":
            n = int(self.read("done_count")[0])
            a = np.empty((n,), dtype=np.int32)
            if n == 0:
                return a
        else:
            raise KeyError(name)
        self._check(self.lib.grl_read_output(self.h, name.encode(), _ptr(a), a.nbytes))
        return a

    def out_ptrs(self):
        o = GrlOutPtrs()
        self._check(self.lib.grl_outputs(self.h, C.byref(o)))
        return o

    def materialize_states(self, first=0, count=None):
        count = self.E - first if count is None else count
        G = self.cfg.grid_size
        a = np.empty((count, 10, G, G, 3), dtype=np.float32)
        self._check(self.lib.grl_swarm_materialize_states(self.h, first, count, _ptr(a), a.nbytes))
        return a

    # -- transforms / returns
    def transform_actions(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        cols = 2 if self.kind == ENV_SWARM else (1 if self.kind == ENV_SOLOW else (4 if self.kind == ENV_TICKER else self.cfg.n_assets))
        flat = a.reshape(-1, cols)
        out = np.empty_like(flat)
        self._check(self.lib.grl_transform_actions_host(self.h, _ptr(flat), _ptr(out), flat.shape[0]))
        return out.reshape(a.shape)

    def returns(self, rewards, values, boot, gamma, masks=None, lam=1.0, scale=1.0, clip=None):
        r = np.ascontiguousarray(rewards, dtype=np.float32)
        v = np.ascontiguousarray(values, dtype=np.float32)
        b = np.ascontiguousarray(boot, dtype=np.float32)
        T, B = r.shape
        m = None if masks is None else np.ascontiguousarray(masks, dtype=np.float32)
        y, adv = np.empty((T, B), np.float32), np.empty((T, B), np.float32)
        lo, hi = (0.0, 0.0) if clip is None else clip
        self._check(self.lib.grl_returns(self.h, _ptr(r), _ptr(v), None if m is None else _ptr(m), _ptr(b), T, B,
                                         gamma, lam, scale, lo, hi, _ptr(y), _ptr(adv)))
        return y, adv

    # -- raw device helpers
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._check(self.lib.grl_dev_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        self._check(self.lib.grl_dev_free(self.h, C.c_void_p(ptr)))

    def dev_upload(self, ptr, arr):
        a = np.ascontiguousarray(arr)
        self._check(self.lib.grl_dev_upload(self.h, C.c_void_p(ptr), _ptr(a), a.nbytes))

    def dev_download(self, ptr, shape, dtype):
        a = np.empty(shape, dtype=dtype)
        self._check(self.lib.grl_dev_download(self.h, _ptr(a), C.c_void_p(ptr), a.nbytes))
        return a

    def dev_randn(self, ptr, n, stream=0, counter=0):
        self._check(self.lib.grl_dev_randn(self.h, C.c_void_p(ptr), n, stream, counter))

    def dev_copy(self, dst, src, nbytes):
        self._check(self.lib.grl_dev_copy(self.h, C.c_void_p(dst), C.c_void_p(src), nbytes))

    def transform_actions_device(self, ptr, rows):
        self._check(self.lib.grl_transform_actions_device(self.h, C.c_void_p(ptr), rows))

    def returns_device(self, r, v, mask, boot, T, B, gamma, lam, scale, clip_lo, clip_hi, y, adv):
        self._check(self.lib.grl_returns_device(self.h, C.c_void_p(r), C.c_void_p(v), C.c_void_p(mask) if mask else None,
                                                C.c_void_p(boot), T, B, gamma, lam, scale, clip_lo, clip_hi,
                                                C.c_void_p(y), C.c_void_p(adv)))

    def profile_enable(self, on=True):
        self._check(self.lib.grl_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        n, ms = C.c_int32(), C.c_float()
        self._check(self.lib.grl_profile_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    # -- R6: per-env episode bookkeeping of the learner loop (paac.py:142-157, 331-349), kept on the device
    def episodes_enable(self, capacity=None):
        self._ep_capacity = int(capacity if capacity is not None else max(1024, 32 * self.E))
        self._check(self.lib.grl_episodes_enable(self.h, self._ep_capacity))

    def episodes_read(self):
        """Finished episodes since the last read, in the reference's append order (step, env): structured array
        (step_index, env, length, total_reward).  Raises if records were dropped (capacity too small for the read interval)."""
        buf = np.zeros(self._ep_capacity, EPISODE_DTYPE)
        n, dropped = C.c_int32(), C.c_int32()
        self._check(self.lib.grl_episodes_read(self.h, _ptr(buf), self._ep_capacity, C.byref(n), C.byref(dropped)))
        if dropped.value:
            raise GrlError(-1, "%d finished episodes were dropped: raise the capacity of episodes_enable()" % dropped.value)
        return buf[:n.value].copy()

    def episodes_running(self):
        tot, ln = np.zeros(self.E, np.float64), np.zeros(self.E, np.int32)
        self._check(self.lib.grl_episodes_running(self.h, _ptr(tot), _ptr(ln)))
        return tot, ln

    def timer_start(self):
        self._check(self.lib.grl_timer_start(self.h))

    def timer_stop(self):
        self._check(self.lib.grl_timer_stop(self.h))

    def timer_ms(self):
        ms = C.c_float()
        self._check(self.lib.grl_timer_ms(self.h, C.byref(ms)))
        return ms.value
