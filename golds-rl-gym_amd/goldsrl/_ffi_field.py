"""ctypes binding of ConvPolicyVFieldNetwork on the device (C ABI: include/goldsrl_fieldnet.h)."""
import ctypes as C

import numpy as np

from . import _ffi

_P, _I, _F = C.c_void_p, C.c_int32, C.c_float


class GrlFieldnetConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32), ("filters", C.c_int32),
                ("conv_layers", C.c_int32), ("num_actions", C.c_int32), ("max_samples", C.c_int32), ("scale", C.c_float),
                ("entropy_beta", C.c_float), ("clip_norm", C.c_float)]


FIELD_SIGNATURES = {
    "grl_fieldnet_config_default": (C.c_int, [C.POINTER(GrlFieldnetConfig)]),
    "grl_fieldnet_create": (C.c_int, [_P, C.POINTER(GrlFieldnetConfig), C.POINTER(_P)]),
    "grl_fieldnet_destroy": (C.c_int, [_P]),
    "grl_fieldnet_last_error": (C.c_char_p, [_P]),
    "grl_fieldnet_num_params": (C.c_int64, [_P]),
    "grl_fieldnet_set_params": (C.c_int, [_P, _P, C.c_int64]),
    "grl_fieldnet_get_params": (C.c_int, [_P, _P, C.c_int64]),
    "grl_fieldnet_get_grads": (C.c_int, [_P, _P, C.c_int64]),
    "grl_fieldnet_predict": (C.c_int, [_P, _I, _P, _P, _P, _P, _P]),
    "grl_fieldnet_train": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _F, _I, _P]),
}


def field_param_shapes(height=32, width=32, channels=3, filters=5, conv_layers=2, num_actions=3, fc_hidden=32):
    """tf.trainable_variables() order of ConvPolicyVFieldNetwork (policy_v_network.py:95-168)."""
    fh, fw = int(height / (2 ** conv_layers)), int(width / (2 ** conv_layers))
    shapes, cin = [], channels
    for i in range(conv_layers):
        shapes += [("conv%d_w" % i, (3, 3, cin, filters)), ("conv%d_b" % i, (filters,))]
        cin = filters
    hwa = height * width * num_actions
    return shapes + [
        ("dense1_w", (fh * fw * filters, 2 * fc_hidden)), ("dense1_b", (2 * fc_hidden,)), ("dense2_w", (2 * fc_hidden, fc_hidden)), ("dense2_b", (fc_hidden,)),
        ("pol1_w", (fc_hidden, 2 * fc_hidden)), ("pol1_b", (2 * fc_hidden,)), ("pol2_w", (2 * fc_hidden, 2 * hwa)), ("pol2_b", (2 * hwa,)),
        ("mu_w", (2 * hwa, hwa)), ("mu_b", (hwa,)), ("sigma_w", (2 * hwa, hwa)), ("sigma_b", (hwa,)),
        ("v1_w", (fc_hidden, 2 * fc_hidden)), ("v1_b", (2 * fc_hidden,)), ("v2_w", (2 * fc_hidden, fc_hidden)), ("v2_b", (fc_hidden,)),
        ("v3_w", (fc_hidden, 1)), ("v3_b", (1,))]


def glorot_uniform_flat(seed=3, **geometry):
    """tf.layers defaults: glorot-uniform kernels, zero biases -> flat float32 vector."""
    rng = np.random.RandomState(seed)
    parts = []
    for name, shape in field_param_shapes(**geometry):
        if name.endswith("_w"):
            if len(shape) == 2:
                fan_in, fan_out = shape
            else:
                rf = int(np.prod(shape[:-2]))
                fan_in, fan_out = rf * shape[-2], rf * shape[-1]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            parts.append(rng.uniform(-lim, lim, size=shape).astype(np.float32).reshape(-1))
        else:
            parts.append(np.zeros(int(np.prod(shape)), np.float32))
    return np.concatenate(parts)


class FieldNet(object):
    def __init__(self, engine, **kw):
        self.lib = _ffi.load_library(extra_signatures=FIELD_SIGNATURES)
        self.eng = engine
        cfg = GrlFieldnetConfig()
        self.lib.grl_fieldnet_config_default(C.byref(cfg))
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown grl_fieldnet_config field %r" % k)
            setattr(cfg, k, v)
        self.cfg = cfg
        n = C.c_void_p()
        rc = self.lib.grl_fieldnet_create(engine.h, C.byref(cfg), C.byref(n))
        if rc != _ffi.OK:
            raise _ffi.GrlError(rc, self.lib.grl_last_error(engine.h).decode())
        self.n = n
        self.num_params = int(self.lib.grl_fieldnet_num_params(n))

    def _check(self, rc):
        if rc != _ffi.OK:
            raise _ffi.GrlError(rc, self.lib.grl_fieldnet_last_error(self.n).decode())

    def close(self):
        if getattr(self, "n", None):
            self.lib.grl_fieldnet_destroy(self.n)
            self.n = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, flat):
        a = np.ascontiguousarray(flat, np.float32)
        self._check(self.lib.grl_fieldnet_set_params(self.n, _ffi._ptr(a), a.size))

    def get_params(self):
        a = np.empty(self.num_params, np.float32)
        self._check(self.lib.grl_fieldnet_get_params(self.n, _ffi._ptr(a), a.size))
        return a

    def get_grads(self):
        a = np.empty(self.num_params, np.float32)
        self._check(self.lib.grl_fieldnet_get_grads(self.n, _ffi._ptr(a), a.size))
        return a

    def _inputs(self, states, positions):
        c = self.cfg
        s = np.ascontiguousarray(states, np.float32)
        p = np.ascontiguousarray(positions, np.int32)
        if s.ndim != 4 or s.shape[1:] != (c.height, c.width, c.channels) or p.shape != (s.shape[0], 2):
            raise ValueError("expected states (n,%d,%d,%d) and positions (n,2), got %s and %s" % (c.height, c.width, c.channels, s.shape, p.shape))
        return s, p

    def predict(self, states, positions):
        s, p = self._inputs(states, positions)
        n, A = s.shape[0], int(self.cfg.num_actions)
        mu, sg, vs = np.empty((n, A), np.float32), np.empty((n, A), np.float32), np.empty(n, np.float32)
        self._check(self.lib.grl_fieldnet_predict(self.n, n, _ffi._ptr(s), _ffi._ptr(p), _ffi._ptr(mu), _ffi._ptr(sg), _ffi._ptr(vs)))
        return {"mu": mu, "sigma": sg, "vs": vs}

    def train(self, states, positions, actions, advantages, critic_target, lr, apply_update=True):
        s, p = self._inputs(states, positions)
        n = s.shape[0]
        a = np.ascontiguousarray(actions, np.float32)
        adv, y = np.ascontiguousarray(advantages, np.float32), np.ascontiguousarray(critic_target, np.float32)
        if a.shape != (n, int(self.cfg.num_actions)) or adv.shape != (n,) or y.shape != (n,):
            raise ValueError("train: actions (n,A), advantages (n,), critic_target (n,) expected")
        stats = np.zeros(4, np.float32)
        self._check(self.lib.grl_fieldnet_train(self.n, n, _ffi._ptr(s), _ffi._ptr(p), _ffi._ptr(a), _ffi._ptr(adv), _ffi._ptr(y), lr,
                                                1 if apply_update else 0, _ffi._ptr(stats)))
        return dict(zip(("loss", "policy_loss", "critic_loss_mean", "global_norm"), stats.tolist()))
