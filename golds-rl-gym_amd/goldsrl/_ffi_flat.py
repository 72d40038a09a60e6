"""ctypes binding of FlatPolicyVNetwork on the device (C ABI: include/goldsrl_flatnet.h)."""
import ctypes as C

import numpy as np

from . import _ffi

_P, _I, _F, _SZ = C.c_void_p, C.c_int32, C.c_float, C.c_size_t


class GrlFnetConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("static_size", C.c_int32), ("temporal_size", C.c_int32), ("rnn_length", C.c_int32),
                ("num_actions", C.c_int32), ("rnn_hidden", C.c_int32), ("static_hidden", C.c_int32), ("max_samples", C.c_int32),
                ("scale", C.c_float), ("clip_norm", C.c_float), ("gamma", C.c_float), ("mu_bound", C.c_float), ("gae_lambda", C.c_float)]


FNET_SIGNATURES = {
    "grl_fnet_config_default": (C.c_int, [C.POINTER(GrlFnetConfig)]),
    "grl_fnet_create": (C.c_int, [_P, C.POINTER(GrlFnetConfig), C.POINTER(_P)]),
    "grl_fnet_destroy": (C.c_int, [_P]),
    "grl_fnet_last_error": (C.c_char_p, [_P]),
    "grl_fnet_num_params": (C.c_int64, [_P]),
    "grl_fnet_set_params": (C.c_int, [_P, _P, C.c_int64]),
    "grl_fnet_get_params": (C.c_int, [_P, _P, C.c_int64]),
    "grl_fnet_get_grads": (C.c_int, [_P, _P, C.c_int64]),
    "grl_fnet_get_action_counter": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "grl_fnet_set_action_counter": (C.c_int, [_P, C.c_uint64]),
    "grl_fnet_get_optimizer_state": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "grl_fnet_set_optimizer_state": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64]),
    "grl_fnet_predict": (C.c_int, [_P, _I, _P, _P, _P, _P, _P]),
    "grl_fnet_predict_env": (C.c_int, [_P, _P, _P, _P]),
    "grl_fnet_train": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _F, _I, _P]),
    "grl_fnet_rollout": (C.c_int, [_P, _I]),
    "grl_fnet_set_keep_activations": (C.c_int, [_P, _I]),
    "grl_fnet_train_rollout": (C.c_int, [_P, _F, _P]),
    "grl_fnet_read_rollout": (C.c_int, [_P, C.c_char_p, _P, _SZ]),
    "grl_fnet_train_rollout_grads": (C.c_int, [_P, _P]),
    "grl_fnet_set_grads": (C.c_int, [_P, _P, C.c_int64]),
    "grl_fnet_apply_grads": (C.c_int, [_P, _F, _F, _P]),
    "grl_fnet_comm_init": (C.c_int, [_P, _P, _SZ, _I, _I]),
    "grl_fnet_comm_broadcast_params": (C.c_int, [_P, _I]),
    "grl_fnet_comm_destroy": (C.c_int, [_P]),
    "grl_fnet_comm_info": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "grl_fnet_rollout_stage_times": (C.c_int, [_P, _P, _I, _P]),
    "grl_comm_unique_id_bytes": (C.c_size_t, []),
    "grl_comm_unique_id": (C.c_int, [_P, _SZ]),
}


def flat_param_shapes(static_size=2, temporal_size=2, num_actions=1, H=32, S=32):
    """tf.trainable_variables() order of FlatPolicyVNetwork (policy_v_network.py:207-244, a3c/estimators.py:18-28)."""
    return [
        ("gru_gates_w", (temporal_size + H, 2 * H)), ("gru_gates_b", (2 * H,)), ("gru_cand_w", (temporal_size + H, H)), ("gru_cand_b", (H,)),
        ("temporal_w", (H, 2 * H)), ("temporal_b", (2 * H,)), ("static1_w", (static_size, 2 * H)), ("static1_b", (2 * H,)),
        ("static2_w", (2 * H, H)), ("static2_b", (H,)),
        ("mu1_w", (3 * H, 2 * S)), ("mu1_b", (2 * S,)), ("mu2_w", (2 * S, S)), ("mu2_b", (S,)), ("mu3_w", (S, num_actions)), ("mu3_b", (num_actions,)),
        ("sig1_w", (3 * H, 2 * S)), ("sig1_b", (2 * S,)), ("sig2_w", (2 * S, S)), ("sig2_b", (S,)), ("sig3_w", (S, num_actions)), ("sig3_b", (num_actions,)),
        ("v1_w", (3 * H, 2 * S)), ("v1_b", (2 * S,)), ("v2_w", (2 * S, 1)), ("v2_b", (1,)),
    ]


def default_init_flat(seed=3, **kw):
    """glorot-uniform kernels, zero biases except GRU gate bias = 1 and sigma-head bias = -1."""
    rng = np.random.RandomState(seed)
    parts = []
    for name, shape in flat_param_shapes(**kw):
        if name.endswith("_w"):
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            parts.append(rng.uniform(-lim, lim, size=shape).reshape(-1))
        elif name == "gru_gates_b":
            parts.append(np.ones(shape))
        elif name == "sig3_b":
            parts.append(-np.ones(shape))
        else:
            parts.append(np.zeros(shape))
    return np.concatenate(parts).astype(np.float32)


class FlatNet(object):
    def __init__(self, engine, **kw):
        self.lib = _ffi.load_library(extra_signatures=FNET_SIGNATURES)
        self.eng = engine
        cfg = GrlFnetConfig()
        self.lib.grl_fnet_config_default(C.byref(cfg))
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown grl_fnet_config field %r" % k)
            setattr(cfg, k, v)
        self.cfg = cfg
        n = C.c_void_p()
        rc = self.lib.grl_fnet_create(engine.h, C.byref(cfg), C.byref(n))
        if rc != _ffi.OK:
            raise _ffi.GrlError(rc, self.lib.grl_last_error(engine.h).decode())
        self.n = n
        self.num_params = int(self.lib.grl_fnet_num_params(n))

    def _check(self, rc):
        if rc != _ffi.OK:
            raise _ffi.GrlError(rc, self.lib.grl_fnet_last_error(self.n).decode())

    def close(self):
        if getattr(self, "n", None):
            self.lib.grl_fnet_destroy(self.n)
            self.n = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, flat):
        a = np.ascontiguousarray(flat, np.float32)
        self._check(self.lib.grl_fnet_set_params(self.n, _ffi._ptr(a), a.size))

    def get_params(self):
        a = np.empty(self.num_params, np.float32)
        self._check(self.lib.grl_fnet_get_params(self.n, _ffi._ptr(a), a.size))
        return a

    def get_grads(self):
        a = np.empty(self.num_params, np.float32)
        self._check(self.lib.grl_fnet_get_grads(self.n, _ffi._ptr(a), a.size))
        return a

    def get_optimizer_state(self):
        """Adam moments and the number of updates applied: with the parameters, the estimator's whole training state."""
        m, v = np.empty(self.num_params, np.float32), np.empty(self.num_params, np.float32)
        step = C.c_int64(0)
        self._check(self.lib.grl_fnet_get_optimizer_state(self.n, _ffi._ptr(m), _ffi._ptr(v), m.size, C.byref(step)))
        return {"adam_m": m, "adam_v": v, "adam_step": int(step.value)}

    def set_optimizer_state(self, adam_m, adam_v, adam_step):
        m, v = np.ascontiguousarray(adam_m, np.float32), np.ascontiguousarray(adam_v, np.float32)
        self._check(self.lib.grl_fnet_set_optimizer_state(self.n, _ffi._ptr(m), _ffi._ptr(v), m.size, int(adam_step)))

    def get_action_counter(self):
        v = C.c_uint64(0)
        self._check(self.lib.grl_fnet_get_action_counter(self.n, C.byref(v)))
        return int(v.value)

    def set_action_counter(self, value):
        self._check(self.lib.grl_fnet_set_action_counter(self.n, int(value)))

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

    def _outs(self, n):
        A = self.cfg.num_actions
        return np.empty((n, A), np.float32), np.empty((n, A), np.float32), np.empty(n, np.float32)

    def predict(self, states, histories):
        s = np.ascontiguousarray(states, np.float32); h = np.ascontiguousarray(histories, np.float32)
        mu, sg, vs = self._outs(s.shape[0])
        self._check(self.lib.grl_fnet_predict(self.n, s.shape[0], _ffi._ptr(s), _ffi._ptr(h), _ffi._ptr(mu), _ffi._ptr(sg), _ffi._ptr(vs)))
        return {"mu": mu, "sigma": sg, "vs": vs}

    def predict_env(self):
        mu, sg, vs = self._outs(self.eng.E)
        self._check(self.lib.grl_fnet_predict_env(self.n, _ffi._ptr(mu), _ffi._ptr(sg), _ffi._ptr(vs)))
        return {"mu": mu, "sigma": sg, "vs": vs}

    def train(self, states, histories, actions, advantages, critic_target, lr, apply_update=True):
        arrs = [np.ascontiguousarray(a, np.float32) for a in (states, histories, actions, advantages, critic_target)]
        stats = np.zeros(4, np.float32)
        self._check(self.lib.grl_fnet_train(self.n, arrs[0].shape[0], *[_ffi._ptr(a) for a in arrs], lr, 1 if apply_update else 0, _ffi._ptr(stats)))
        return dict(zip(("loss", "policy_loss", "critic_loss_mean", "global_norm"), stats.tolist()))

    def rollout(self, T):
        self._check(self.lib.grl_fnet_rollout(self.n, T))

    def set_keep_activations(self, on):
        """The rollouts that follow fill the training workspace; train_rollout on them starts at the backward pass (bit-identical
        gradients, the rollout pays the stores).  Off by default."""
        self._check(self.lib.grl_fnet_set_keep_activations(self.n, 1 if on else 0))

    def train_rollout(self, lr):
        stats = np.zeros(4, np.float32)
        self._check(self.lib.grl_fnet_train_rollout(self.n, lr, _ffi._ptr(stats)))
        return dict(zip(("loss", "policy_loss", "critic_loss_mean", "global_norm"), stats.tolist()))

    def train_rollout_grads(self):
        """Loss + backward over the last rollout only: the local mean gradient stays in the net (get_grads)."""
        stats = np.zeros(4, np.float32)
        self._check(self.lib.grl_fnet_train_rollout_grads(self.n, _ffi._ptr(stats)))
        return dict(zip(("loss", "policy_loss", "critic_loss_mean", "global_norm"), stats.tolist()))

    def set_grads(self, flat):
        a = np.ascontiguousarray(flat, np.float32)
        self._check(self.lib.grl_fnet_set_grads(self.n, _ffi._ptr(a), a.size))

    def apply_grads(self, lr, grad_scale=1.0):
        """clip_by_global_norm(grad_scale * grads) + Adam(lr) on the gradient currently in the net."""
        stats = np.zeros(4, np.float32)
        self._check(self.lib.grl_fnet_apply_grads(self.n, lr, grad_scale, _ffi._ptr(stats)))
        return dict(zip(("loss", "policy_loss", "critic_loss_mean", "global_norm"), stats.tolist()))

    # ---- multi-GPU: one RCCL all-reduce of the flat gradient per rollout (include/goldsrl_flatnet.h)
    def comm_unique_id(self):
        n = int(self.lib.grl_comm_unique_id_bytes())
        buf = np.zeros(n, np.uint8)
        rc = self.lib.grl_comm_unique_id(_ffi._ptr(buf), n)
        if rc != _ffi.OK:
            raise _ffi.GrlError(rc, "grl_comm_unique_id")
        return buf

    def comm_init(self, unique_id, rank, world_size):
        buf = np.ascontiguousarray(unique_id, np.uint8)
        self._check(self.lib.grl_fnet_comm_init(self.n, _ffi._ptr(buf), buf.size, rank, world_size))

    def comm_broadcast_params(self, root=0):
        self._check(self.lib.grl_fnet_comm_broadcast_params(self.n, root))

    def rollout_stage_times(self):
        """Constant-clock ticks (10 ns) of workgroup 0 at every barrier of the last persistent rollout (first call: attaches)."""
        buf = np.zeros(4096, np.int64)
        n = C.c_int32()
        self._check(self.lib.grl_fnet_rollout_stage_times(self.n, _ffi._ptr(buf), 4096, C.byref(n)))
        return buf[:n.value].copy()

    def comm_info(self):
        """What RCCL reports for the attached communicator (ranks = ncclCommCount, 0 without one) and the all-reduce timing."""
        cnt, ur, calls, tot, last = C.c_int32(), C.c_int32(), C.c_int64(), C.c_double(), C.c_float()
        self._check(self.lib.grl_fnet_comm_info(self.n, C.byref(cnt), C.byref(ur), C.byref(calls), C.byref(tot), C.byref(last)))
        return {"rccl_ranks": cnt.value, "rccl_user_rank": ur.value, "allreduce_calls": calls.value,
                "allreduce_ms_total": tot.value, "allreduce_ms_last": last.value}

    def comm_destroy(self):
        self._check(self.lib.grl_fnet_comm_destroy(self.n))

    def read_rollout(self, which, shape):
        a = np.empty(shape, np.float32)
        self._check(self.lib.grl_fnet_read_rollout(self.n, which.encode(), _ffi._ptr(a), a.nbytes))
        return a
