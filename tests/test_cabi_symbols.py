"""CPU: the C-ABI library builds, loads, and exports every symbol include/*.h declares; without a GPU
the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("goldsrl.h", "goldsrl_net.h", "goldsrl_flatnet.h", "goldsrl_fieldnet.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(grl_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_every_declared_symbol_is_exported():
    from goldsrl import _ffi, _ffi_field, _ffi_flat, _ffi_net
    lib = _ffi.load_library(extra_signatures=dict(_ffi_net.NET_SIGNATURES, **dict(_ffi_flat.FNET_SIGNATURES, **_ffi_field.FIELD_SIGNATURES)))
    declared = _declared()
    assert len(declared) >= 50
    for name in declared:
        assert hasattr(lib, name), "include/*.h declares %s but libgoldsrl.so does not export it" % name
    bound = set(_ffi.SIGNATURES) | set(_ffi_net.NET_SIGNATURES) | set(_ffi_flat.FNET_SIGNATURES) | set(_ffi_field.FIELD_SIGNATURES)
    assert bound == set(declared), (sorted(bound - set(declared)), sorted(set(declared) - bound))
    assert lib.grl_abi_version() == 1


def test_config_struct_matches_header_and_defaults():
    from goldsrl import _ffi
    lib = _ffi.load_library()
    cfg = _ffi.GrlConfig()
    assert lib.grl_config_default(_ffi.ENV_SWARM, ctypes.byref(cfg)) == 0
    assert cfg.struct_size == ctypes.sizeof(_ffi.GrlConfig)          # the library's sizeof(grl_config)
    assert (cfg.max_episode_steps, cfg.grid_size, cfg.num_envs) == (128, 84, 32)
    assert lib.grl_config_default(_ffi.ENV_SOLOW, ctypes.byref(cfg)) == 0
    assert (cfg.max_episode_steps, cfg.solow_p, cfg.solow_q, cfg.solow_tape_len, cfg.rnn_length) == (1024, 1, 1, 2048, 5)
    assert lib.grl_config_default(7, ctypes.byref(cfg)) == _ffi.E_INVALID


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_has_gpu(), reason="this check is about a GPU-less host")
def test_no_gpu_means_loud_failure_not_a_fallback():
    from goldsrl import _ffi
    with pytest.raises(_ffi.GrlError) as ei:
        _ffi.Engine(_ffi.ENV_SWARM, 4)
    assert ei.value.code == _ffi.E_NO_DEVICE and "no CPU path" in str(ei.value)
    from goldsrl.envs import multiagent
    with pytest.raises(_ffi.GrlError):
        multiagent.SwarmEnv()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "golds-rl-gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f
