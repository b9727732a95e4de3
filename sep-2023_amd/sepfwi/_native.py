"""Loader / builder of libsepfwi.so, the HIP propagator behind the C ABI of include/sepfwi.h.

There is no CPU fallback: if the library cannot be loaded every operator call raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)                      # sep-2023_amd/
CSRC = os.path.join(_ROOT, "csrc")
LIB_PATH = os.path.join(_ROOT, "libsepfwi.so")
# The same sources built with -DSEPFWI_PROBES: sepfwi_set_option then also knows the tuning knobs (tile shapes, wave counts, launch
# structures) and the timing-only switches that the shipped library does not expose (include/sepfwi.h).  Used by scripts/ab_bench.py
# and by the tests that prove every selectable kernel structure bit-identical; never by the operators unless asked to (use_variant).
PROBES_LIB_PATH = os.path.join(_ROOT, "libsepfwi_probes.so")
# ... and with -DSEPFWI_PK_FAULT=<tile>: a fault-injection build whose persistent loop stalls on purpose (one tile stops publishing), for
# the one test that must see the loop's time-out path work (tests/test_gpu_parity.py::test_loop_failure_path_reports_and_recovers).
FAULT_LIB_PATH = os.path.join(_ROOT, "libsepfwi_fault.so")
VARIANTS = {"default": (LIB_PATH, []), "probes": (PROBES_LIB_PATH, ["-DSEPFWI_PROBES"]), "fault": (FAULT_LIB_PATH, ["-DSEPFWI_PK_FAULT=17"])}
PUBLIC_OPTIONS = ("bwd_fuse", "batch", "img_every", "quiet_skip", "obs_cache_mb", "probe")
SOURCES = ["kernels.hip", "param_maps.hip", "conditioning.hip", "session.cpp", "session_run.cpp", "session_persist.cpp", "session_batched.cpp", "obs_store.cpp", "host_checks.cpp", "persist_plan.cpp", "inject_plan.cpp", "config.cpp", "capi.cpp"]
HEADERS = ["kernels.hpp", "kernels_device.hpp", "kernels_bodies.hpp", "kernels_quiet.hpp", "kernels_step.hpp", "kernels_persist.hpp", "kernels_aux.hpp", "param_maps.hpp", "conditioning.hpp", "device_common.hpp", "device_alloc.hpp", "obs_store.hpp", "host_checks.hpp", "persist_plan.hpp", "inject_plan.hpp", "errors.hpp", "hip_check.hpp", "session.hpp", "config.hpp", "fwi_types.hpp", "json_min.hpp",
           os.path.join("..", "..", "include", "sepfwi.h")]

_libs = {}
_active = os.environ.get("SEPFWI_LIB_VARIANT", "default")


class Stats(C.Structure):
    """mirror of struct sepfwi_stats (include/sepfwi.h)"""
    _fields_ = [("fwd_ms", C.c_double), ("bwd_ms", C.c_double), ("total_ms", C.c_double),
                ("cell_updates", C.c_double), ("fwd_steps", C.c_longlong), ("bwd_steps", C.c_longlong),
                ("launches", C.c_longlong), ("device_bytes", C.c_longlong), ("n_c", C.c_int),
                ("probe_kernel_us", C.c_double), ("probe_calls", C.c_longlong),
                ("obs_device_bytes", C.c_longlong), ("obs_host_bytes", C.c_longlong), ("obs_evictions", C.c_longlong), ("persist_steps", C.c_longlong), ("quiet_active", C.c_longlong), ("quiet_total", C.c_longlong)]


def needs_build(variant: str = "default") -> bool:
    path = VARIANTS[variant][0]
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, variant: str = "all") -> str:
    """hipcc --offload-arch=gfx950 ... (cross-compiles without a GPU; ~30 s per variant)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for name in (VARIANTS if variant == "all" else (variant,)):
        path, flags = VARIANTS[name]
        if not (force or needs_build(name)):
            continue
        # -ffp-contract=off: no fused multiply-adds chosen per kernel by the compiler, so every kernel structure (stream, batched,
        # unfused, persistent) gives bit-identical results; the kernels are memory-bound, it costs nothing
        extra = os.environ.get("SEPFWI_HIPCC_FLAGS", "").split()   # experiments only (e.g. -fgpu-flush-denormals-to-zero)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall"] + flags + extra + ["-o", path] + SOURCES + ["-ldl"]   # hipFFT is opened lazily (csrc/conditioning.hip)
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


@contextlib.contextmanager
def use_variant(name: str):
    """Route every operator call of the block to another build of the library (its own sessions and option block)."""
    global _active
    prev, _active = _active, name
    try:
        yield lib()
    finally:
        _active = prev


def lib():
    """The loaded library (ctypes.CDLL) with argument types declared."""
    if _active in _libs:
        return _libs[_active]
    path = VARIANTS[_active][0]
    # torch first: both link libamdhip64.so.7 and must share ONE HIP runtime in the process.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is not built (%s missing). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback for the propagator." % (os.path.basename(path), path))
    L = C.CDLL(path)
    fp, ip = C.c_void_p, C.c_void_p
    L.sepfwi_last_error.restype = C.c_char_p
    L.sepfwi_cufd.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, ip, C.c_char_p]
    L.sepfwi_cufd_stream.argtypes = L.sepfwi_cufd.argtypes + [C.c_void_p, C.c_int]
    L.sepfwi_cpml_profiles.argtypes = [fp] * 6 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    L.sepfwi_stf_taper.argtypes = [fp, C.c_int, C.c_float, C.c_float]
    L.sepfwi_shot_split.argtypes = [C.c_int, C.c_int, ip]
    L.sepfwi_get_stats.argtypes = [C.c_char_p, C.c_int, C.POINTER(Stats)]
    L.sepfwi_set_option.argtypes = [C.c_char_p, C.c_int]
    L.sepfwi_get_option.argtypes = [C.c_char_p]
    L.sepfwi_param_forward.argtypes = [C.c_int] * 5 + [fp] * 10 + [C.c_void_p]
    L.sepfwi_param_backward.argtypes = [C.c_int] * 5 + [fp] * 13 + [C.c_void_p]
    L.sepfwi_debug_field.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, fp]
    L.sepfwi_loop_status.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.sepfwi_set_observed.argtypes = [C.c_char_p, C.c_int, C.c_int, fp, C.c_int, C.c_int]
    for f in ("sepfwi_cufd", "sepfwi_cufd_stream", "sepfwi_cpml_profiles", "sepfwi_stf_taper", "sepfwi_shot_split",
              "sepfwi_get_stats", "sepfwi_loop_status", "sepfwi_set_option", "sepfwi_get_option", "sepfwi_debug_field", "sepfwi_set_observed",
              "sepfwi_param_forward", "sepfwi_param_backward", "sepfwi_version", "sepfwi_device_count"):
        getattr(L, f).restype = C.c_int
    L.sepfwi_release_all.restype = None
    L.sepfwi_invalidate_observed.restype = None
    _libs[_active] = L
    return L


EXPORTS = ["sepfwi_last_error", "sepfwi_version", "sepfwi_device_count", "sepfwi_cufd", "sepfwi_cufd_stream",
           "sepfwi_release_all", "sepfwi_invalidate_observed", "sepfwi_cpml_profiles", "sepfwi_stf_taper",
           "sepfwi_shot_split", "sepfwi_get_stats", "sepfwi_loop_status", "sepfwi_set_option", "sepfwi_get_option", "sepfwi_debug_field",
           "sepfwi_set_observed", "sepfwi_param_forward", "sepfwi_param_backward"]


class SepFwiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sepfwi error %d: %s" % (code, msg))
        self.code = code


def check(rc: int):
    if rc != 0:
        raise SepFwiError(rc, lib().sepfwi_last_error().decode(errors="replace"))
