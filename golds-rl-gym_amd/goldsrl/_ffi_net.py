"""ctypes binding of the estimator side of libgoldsrl.so (C ABI: include/goldsrl_net.h)."""
import ctypes as C

import numpy as np

from . import _ffi

_P, _I, _F, _SZ = C.c_void_p, C.c_int32, C.c_float, C.c_size_t


class GrlNetConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("kind", C.c_int32), ("max_chunk_samples", C.c_int32), ("reserved", C.c_int32),
                ("scale", C.c_float), ("entropy_beta", C.c_float), ("clip_norm", C.c_float), ("gamma", C.c_float),
                ("num_actions", C.c_int32), ("reserved2", C.c_int32)]


NET_SIGNATURES = {
    "grl_net_config_default": (C.c_int, [_I, C.POINTER(GrlNetConfig)]),
    "grl_net_create": (C.c_int, [_P, C.POINTER(GrlNetConfig), C.POINTER(_P)]),
    "grl_net_destroy": (C.c_int, [_P]),
    "grl_net_last_error": (C.c_char_p, [_P]),
    "grl_net_num_params": (C.c_int64, [_P]),
    "grl_net_set_params": (C.c_int, [_P, _P, C.c_int64]),
    "grl_net_get_params": (C.c_int, [_P, _P, C.c_int64]),
    "grl_net_get_grads": (C.c_int, [_P, _P, C.c_int64]),
    "grl_net_get_action_counter": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "grl_net_set_action_counter": (C.c_int, [_P, C.c_uint64]),
    "grl_net_get_optimizer_state": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "grl_net_set_optimizer_state": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64]),
    "grl_net_predict": (C.c_int, [_P, _P, _P, _P]),
    "grl_net_predict_obs": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P]),
    "grl_net_rollout": (C.c_int, [_P, _I, _I]),
    "grl_net_train_rollout": (C.c_int, [_P, _F, _P]),
    "grl_net_train_rollout_grads": (C.c_int, [_P, _P]),
    "grl_net_set_grads": (C.c_int, [_P, _P, C.c_int64]),
    "grl_net_apply_grads": (C.c_int, [_P, _F, _F, _P]),
    "grl_net_train_obs": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "grl_net_read_rollout": (C.c_int, [_P, C.c_char_p, _P, _SZ]),
    "grl_net_read_activation": (C.c_int, [_P, C.c_char_p, _P, _SZ]),
    "grl_comm_unique_id_bytes": (C.c_size_t, []),
    "grl_comm_unique_id": (C.c_int, [_P, _SZ]),
    "grl_net_comm_init": (C.c_int, [_P, _P, _SZ, _I, _I]),
    "grl_net_comm_broadcast_params": (C.c_int, [_P, _I]),
    "grl_net_comm_destroy": (C.c_int, [_P]),
    "grl_net_comm_info": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "grl_net_range_info": (C.c_int, [_P, _P, _P, _P]),
    "grl_net_host_times": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "grl_net_set_gemm_f32": (C.c_int, [_P, C.c_int32]),
    "grl_net_range_return_info": (C.c_int, [_P, _P, _P, _P, _P]),
    "grl_net_set_range_return": (C.c_int, [_P, C.c_int32]),
    "grl_net_profile_enable": (C.c_int, [_P, _I]),
    "grl_net_profile_read": (C.c_int, [_P, C.POINTER(_I), C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "grl_net_profile_read_tags": (C.c_int, [_P, _I, _P, _P, _P]),
}

NET_CONV_SINGLE_AGENT = 0

# (name, shape) in flat-vector order == tf.trainable_variables() creation order
# (reference fed_gym/agents/paac/policy_v_network.py:14-59)
def conv_param_shapes(num_actions=2):
    A = int(num_actions)
    return [
        ("conv1_w", (8, 8, 3, 32)), ("conv1_b", (32,)), ("conv2_w", (4, 4, 32, 64)), ("conv2_b", (64,)),
        ("conv3_w", (3, 3, 64, 64)), ("conv3_b", (64,)), ("dense1_w", (3136, 512)), ("dense1_b", (512,)),
        ("dense2_w", (512, 256)), ("dense2_b", (256,)), ("pol1_w", (256, 512)), ("pol1_b", (512,)),
        ("mu_w", (512, A)), ("mu_b", (A,)), ("sigma_w", (512, A)), ("sigma_b", (A,)),
        ("v1_w", (256, 512)), ("v1_b", (512,)), ("v2_w", (512, 256)), ("v2_b", (256,)), ("v3_w", (256, 1)), ("v3_b", (1,)),
    ]


CONV_PARAM_SHAPES = conv_param_shapes(2)      # train_paac_conv.py: SwarmEnvironmentCreator.num_actions = 2


def glorot_uniform_flat(seed=3, num_actions=2):
    """tf.layers defaults: glorot-uniform kernels, zero biases -> flat float32 vector."""
    rng = np.random.RandomState(seed)
    parts = []
    for name, shape in conv_param_shapes(num_actions):
        if name.endswith("_w"):
            if len(shape) == 2:
                fan_in, fan_out = shape
            else:
                rf = int(np.prod(shape[:-2]))
                fan_in, fan_out = rf * shape[-2], rf * shape[-1]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            parts.append(rng.uniform(-lim, lim, size=shape).reshape(-1))
        else:
            parts.append(np.zeros(int(np.prod(shape))))
    return np.concatenate(parts).astype(np.float32)


class ConvNet(object):
    """ConvSingleAgentPolicyNetwork on the device of a Swarm Engine."""

    def __init__(self, engine, **kw):
        self.lib = _ffi.load_library(extra_signatures=NET_SIGNATURES)
        self.eng = engine
        cfg = GrlNetConfig()
        rc = self.lib.grl_net_config_default(NET_CONV_SINGLE_AGENT, C.byref(cfg))
        if rc != _ffi.OK:
            raise _ffi.GrlError(rc, "grl_net_config_default")
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown grl_net_config field %r" % k)
            setattr(cfg, k, v)
        self.cfg = cfg
        n = C.c_void_p()
        rc = self.lib.grl_net_create(engine.h, C.byref(cfg), C.byref(n))
        if rc != _ffi.OK:
            raise _ffi.GrlError(rc, self.lib.grl_last_error(engine.h).decode())
        self.n = n
        self.num_params = int(self.lib.grl_net_num_params(n))

    def _check(self, rc):
        if rc != _ffi.OK:
            raise _ffi.GrlError(rc, self.lib.grl_net_last_error(self.n).decode())

    def close(self):
        if getattr(self, "n", None):
            self.lib.grl_net_destroy(self.n)
            self.n = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, flat):
        a = np.ascontiguousarray(flat, dtype=np.float32)
        self._check(self.lib.grl_net_set_params(self.n, _ffi._ptr(a), a.size))

    def get_params(self):
        a = np.empty(self.num_params, np.float32)
        self._check(self.lib.grl_net_get_params(self.n, _ffi._ptr(a), a.size))
        return a

    def get_grads(self):
        a = np.empty(self.num_params, np.float32)
        self._check(self.lib.grl_net_get_grads(self.n, _ffi._ptr(a), a.size))
        return a

    def get_optimizer_state(self):
        """Adam moments and the number of updates applied: with the parameters, the estimator's whole training state."""
        m, v = np.empty(self.num_params, np.float32), np.empty(self.num_params, np.float32)
        step = C.c_int64(0)
        self._check(self.lib.grl_net_get_optimizer_state(self.n, _ffi._ptr(m), _ffi._ptr(v), m.size, C.byref(step)))
        return {"adam_m": m, "adam_v": v, "adam_step": int(step.value)}

    def set_optimizer_state(self, adam_m, adam_v, adam_step):
        m, v = np.ascontiguousarray(adam_m, np.float32), np.ascontiguousarray(adam_v, np.float32)
        self._check(self.lib.grl_net_set_optimizer_state(self.n, _ffi._ptr(m), _ffi._ptr(v), m.size, int(adam_step)))

    def get_action_counter(self):
        v = C.c_uint64(0)
        self._check(self.lib.grl_net_get_action_counter(self.n, C.byref(v)))
        return int(v.value)

    def set_action_counter(self, value):
        self._check(self.lib.grl_net_set_action_counter(self.n, int(value)))

    def save_checkpoint(self, path, **extra):
        """Flat-weights checkpoint (.npz): parameters in tf.trainable_variables() order, Adam state, caller's scalars."""
        st = self.get_optimizer_state()
        np.savez(path, params=self.get_params(), adam_m=st["adam_m"], adam_v=st["adam_v"], adam_step=st["adam_step"], action_counter=self.get_action_counter(),
                 **{k: np.asarray(v) for k, v in extra.items()})

    def load_checkpoint(self, path):
        with np.load(path) as z:
            self.set_params(z["params"])
            self.set_optimizer_state(z["adam_m"], z["adam_v"], int(z["adam_step"]))
            if "action_counter" in z.files:      # the action-noise stream continues where the saved run stopped
                self.set_action_counter(int(z["action_counter"]))
            return {k: z[k] for k in z.files if k not in ("params", "adam_m", "adam_v", "adam_step", "action_counter")}

    def predict(self):
        B, A = self.eng.E * 10, int(self.cfg.num_actions)
        mu, sg, vs = np.empty((B, A), np.float32), np.empty((B, A), np.float32), np.empty(B, np.float32)
        self._check(self.lib.grl_net_predict(self.n, _ffi._ptr(mu), _ffi._ptr(sg), _ffi._ptr(vs)))
        return {"mu": mu, "sigma": sg, "vs": vs}

    def predict_obs(self, locust_bins, agent_bins, positions):
        lb = np.ascontiguousarray(locust_bins, np.uint8); ab = np.ascontiguousarray(agent_bins, np.uint8)
        ps = np.ascontiguousarray(positions, np.uint8)
        ne = lb.shape[0]
        B, A = ne * 10, int(self.cfg.num_actions)
        mu, sg, vs = np.empty((B, A), np.float32), np.empty((B, A), np.float32), np.empty(B, np.float32)
        self._check(self.lib.grl_net_predict_obs(self.n, ne, _ffi._ptr(lb), _ffi._ptr(ab), _ffi._ptr(ps), _ffi._ptr(mu),
                                                 _ffi._ptr(sg), _ffi._ptr(vs)))
        return {"mu": mu, "sigma": sg, "vs": vs}

    def train_obs(self, locust_bins, agent_bins, positions, actions, advantages, critic_target, lr, apply_update=True):
        lb = np.ascontiguousarray(locust_bins, np.uint8); ab = np.ascontiguousarray(agent_bins, np.uint8)
        ps = np.ascontiguousarray(positions, np.uint8)
        a = np.ascontiguousarray(actions, np.float32); adv = np.ascontiguousarray(advantages, np.float32)
        y = np.ascontiguousarray(critic_target, np.float32)
        if a.shape != (lb.shape[0] * 10, int(self.cfg.num_actions)):
            raise ValueError("train_obs: actions must be (%d, %d), got %s" % (lb.shape[0] * 10, self.cfg.num_actions, a.shape))
        stats = np.zeros(4, np.float32)
        self._check(self.lib.grl_net_train_obs(self.n, lb.shape[0], _ffi._ptr(lb), _ffi._ptr(ab), _ffi._ptr(ps), _ffi._ptr(a),
                                               _ffi._ptr(adv), _ffi._ptr(y), lr, 1 if apply_update else 0, _ffi._ptr(stats)))
        return dict(zip(("loss", "policy_loss", "critic_loss_mean", "global_norm"), stats.tolist()))

    def rollout(self, T, reward_layout=0):
        self._check(self.lib.grl_net_rollout(self.n, T, reward_layout))

    def train_rollout(self, lr):
        stats = np.zeros(4, np.float32)
        self._check(self.lib.grl_net_train_rollout(self.n, lr, _ffi._ptr(stats)))
        return dict(zip(("loss", "policy_loss", "critic_loss_mean", "global_norm"), stats.tolist()))

    def train_rollout_grads(self):
        """Loss + backward over the last rollout only: the local mean gradient stays in the net (get_grads)."""
        stats = np.zeros(4, np.float32)
        self._check(self.lib.grl_net_train_rollout_grads(self.n, _ffi._ptr(stats)))
        return dict(zip(("loss", "policy_loss", "critic_loss_mean", "global_norm"), stats.tolist()))

    def set_grads(self, flat):
        a = np.ascontiguousarray(flat, dtype=np.float32)
        self._check(self.lib.grl_net_set_grads(self.n, _ffi._ptr(a), a.size))

    def apply_grads(self, lr, grad_scale=1.0):
        """clip_by_global_norm(grad_scale * grads) + Adam(lr) on the gradient currently in the net."""
        stats = np.zeros(4, np.float32)
        self._check(self.lib.grl_net_apply_grads(self.n, lr, grad_scale, _ffi._ptr(stats)))
        return dict(zip(("loss", "policy_loss", "critic_loss_mean", "global_norm"), stats.tolist()))

    def read_rollout(self, which, shape, dtype=np.float32):
        a = np.empty(shape, dtype)
        self._check(self.lib.grl_net_read_rollout(self.n, which.encode(), _ffi._ptr(a), a.nbytes))
        return a

    def read_activation(self, which, shape):
        a = np.empty(shape, np.float32)
        self._check(self.lib.grl_net_read_activation(self.n, which.encode(), _ffi._ptr(a), a.nbytes))
        return a

    # -- multi-GPU (RCCL): rank 0 makes the id, everybody attaches
    def comm_unique_id(self):
        n = int(self.lib.grl_comm_unique_id_bytes())
        buf = np.zeros(n, np.uint8)
        rc = self.lib.grl_comm_unique_id(_ffi._ptr(buf), n)
        if rc != _ffi.OK:
            raise _ffi.GrlError(rc, "grl_comm_unique_id")
        return buf

    def comm_init(self, unique_id, rank, world_size):
        buf = np.ascontiguousarray(unique_id, np.uint8)
        self._check(self.lib.grl_net_comm_init(self.n, _ffi._ptr(buf), buf.size, rank, world_size))

    def comm_broadcast_params(self, root=0):
        self._check(self.lib.grl_net_comm_broadcast_params(self.n, root))

    def comm_info(self):
        """What RCCL reports for the attached communicator (ranks = ncclCommCount, 0 without one) and the all-reduce timing."""
        cnt, ur, calls, tot, last = C.c_int32(), C.c_int32(), C.c_int64(), C.c_double(), C.c_float()
        self._check(self.lib.grl_net_comm_info(self.n, C.byref(cnt), C.byref(ur), C.byref(calls), C.byref(tot), C.byref(last)))
        return {"rccl_ranks": cnt.value, "rccl_user_rank": ur.value, "allreduce_calls": calls.value,
                "allreduce_ms_total": tot.value, "allreduce_ms_last": last.value}

    def comm_destroy(self):
        self._check(self.lib.grl_net_comm_destroy(self.n))

    def host_times(self):
        """Wall-clock ms this thread spent enqueueing rollouts / gradient steps and waiting for the device (grl_net_host_times)."""
        ro, up = C.c_int64(0), C.c_int64(0)
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        self._check(self.lib.grl_net_host_times(self.n, C.byref(ro), C.byref(up), C.byref(a), C.byref(b), C.byref(c)))
        return {"rollouts": int(ro.value), "updates": int(up.value), "rollout_enqueue_ms": a.value, "train_enqueue_ms": b.value,
                "train_wait_ms": c.value}

    def range_info(self):
        """Arithmetic form of the GEMMs: 'gemm_f32' once a range violation (or set_gemm_f32) moved the net to the fp32 form,
        'fallbacks' = how often that happened, 'update_skipped' = the last train_rollout* call gave its update up."""
        f32, fb, sk = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.grl_net_range_info(self.n, C.byref(f32), C.byref(fb), C.byref(sk)))
        return {"gemm_f32": bool(f32.value), "fallbacks": fb.value, "update_skipped": bool(sk.value)}

    def set_gemm_f32(self, on=True):
        self._check(self.lib.grl_net_set_gemm_f32(self.n, 1 if on else 0))

    def range_return_info(self):
        """The way back from a range fallback: 'returns' = how often the net went back to the fp16 form, 'clean_passes' = clean updates
        counted so far on the fp32 form, 'needed' = how many in a row it takes (0: never), 'absmax_last' = the largest |GEMM output|
        the last decision saw."""
        r, c, k, a = C.c_int32(), C.c_int32(), C.c_int32(), C.c_float()
        self._check(self.lib.grl_net_range_return_info(self.n, C.byref(r), C.byref(c), C.byref(k), C.byref(a)))
        return {"returns": r.value, "clean_passes": c.value, "needed": k.value, "absmax_last": a.value}

    def set_range_return(self, clean_passes):
        self._check(self.lib.grl_net_set_range_return(self.n, int(clean_passes)))

    def profile_enable(self, on=True):
        self._check(self.lib.grl_net_profile_enable(self.n, 1 if on else 0))

    PROFILE_TAGS = ("other", "dense_small_fwd", "dense_small_dgrad", "dense_small_wgrad", "dense1_patch_fwd", "dense1_patch_dgrad",
                    "dense1_patch_wgrad", "per_env_fwd", "per_env_dgrad", "per_env_wgrad", "slot_products_fwd", "slot_dgrad", "slot_wgrad",
                    "conv2_class_corrections", "per_agent_mode", "-")

    def profile_read_tags(self):
        """{family: (launches, ms, flops)} of the bracketed GEMM launches since profile_enable(True)."""
        k = len(self.PROFILE_TAGS)
        n, ms, fl = np.zeros(k, np.int32), np.zeros(k, np.float32), np.zeros(k, np.float64)
        self._check(self.lib.grl_net_profile_read_tags(self.n, k, _ffi._ptr(n), _ffi._ptr(ms), _ffi._ptr(fl)))
        return {name: (int(n[i]), float(ms[i]), float(fl[i])) for i, name in enumerate(self.PROFILE_TAGS) if n[i]}

    def profile_read(self):
        n, ms, fl = C.c_int32(), C.c_float(), C.c_double()
        self._check(self.lib.grl_net_profile_read(self.n, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value
