"""ctypes binding of the C ABI in include/dronesim.h (libdronesim.so, HIP/gfx950).

There is NO CPU fallback: if the library is missing, import of this module's
`lib()` fails loudly with build instructions.  All pointers handed to the
library are raw device addresses (`tensor.data_ptr()`); PyTorch only provides
the memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# DRONESIM_LIB lets a developer A/B an alternative build of the SAME HIP library (no other backend exists)
LIB_PATH = os.environ.get("DRONESIM_LIB") or os.path.join(_PKG, "libdronesim.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "dronesim.h")
# the float64 verification variant (test infrastructure) is a library of its own: the product library exports the product only
VERIFY_LIB_PATH = os.environ.get("DRONESIM_VERIFY_LIB") or os.path.join(_PKG, "libdronesim_verify.so")
VERIFY_HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "dronesim_verify.h")
VERIFY_SYMBOLS = ("dronesim_step_f64", "dronesim_observe_f64", "dronesim_verify_last_error")

OK, EINVAL, EUNSUPPORTED, ELAUNCH = 0, -1, -2, -3
CONTROL_PROPORTIONAL, CONTROL_GRADIENT = 0, 1
MAX_K = 8
MAX_AGENTS = 1024

# every symbol include/dronesim.h declares (tests check the .so exports all of them)
STATS_SCRATCH_DOUBLES = 769          # DRONESIM_STATS_SCRATCH_DOUBLES (include/dronesim.h)
EPISODE_REDUCE_DOUBLES = 8           # DRONESIM_EPISODE_REDUCE_DOUBLES
SYMBOLS = ("dronesim_step", "dronesim_observe", "dronesim_reset", "dronesim_rollout", "dronesim_control", "dronesim_returns", "dronesim_advantage", "dronesim_episode_stats", "dronesim_mlp_forward", "dronesim_mlp_forward_bf16",
           "dronesim_step_ex", "dronesim_step_call", "dronesim_rollout_ex", "dronesim_rollout_random", "dronesim_reset_ex", "dronesim_reset_observe", "dronesim_episode_reduce",
           "dronesim_mlp_forward_bf16x3", "dronesim_mlp_forward_f16x2", "dronesim_mlp_bf16x3_stages", "dronesim_mlp_rt_blocks", "dronesim_mlp_rt16_blocks", "dronesim_mlp_forward_f16x2_rt",
           "dronesim_last_error", "dronesim_error_string", "dronesim_version")


class DroneParams(C.Structure):
    """Mirror of `struct DroneParams` (include/dronesim.h)."""
    _fields_ = [("N", C.c_int32), ("k", C.c_int32), ("c", C.c_int32), ("max_steps", C.c_int32),
                ("dt", C.c_float), ("q", C.c_float), ("b", C.c_float),
                ("done_radius", C.c_float), ("ghost_factor", C.c_float),
                ("d_hat_min", C.c_float), ("d_hat_max", C.c_float), ("delta_min", C.c_float), ("delta_max", C.c_float),
                ("radius_min", C.c_float), ("radius_max", C.c_float),
                ("xF", C.c_void_p), ("d_hat", C.c_void_p), ("delta", C.c_void_p),
                ("radius", C.c_void_p), ("xF_lo", C.c_void_p)]


class DroneEpisodeAcc(C.Structure):
    """Mirror of `struct DroneEpisodeAcc` (include/dronesim.h): one 64-byte record per env."""
    _fields_ = [("ep_return", C.c_double), ("ep_true_return", C.c_double),
                ("ep_collisions", C.c_int32), ("ep_len", C.c_int32), ("episodes", C.c_int32), ("reserved", C.c_int32),
                ("done_return", C.c_double), ("done_true_return", C.c_double),
                ("done_collisions", C.c_int64), ("done_len", C.c_int64)]


class DroneEpisodeCtl(C.Structure):
    """Mirror of `struct DroneEpisodeCtl` (include/dronesim.h)."""
    _fields_ = [("acc", C.c_void_p), ("auto_reset", C.c_int32), ("div_x", C.c_int32), ("div_y", C.c_int32),
                ("pitch", C.c_float), ("seed", C.c_uint64), ("env_base", C.c_int64), ("episode", C.c_void_p),
                ("z_final", C.c_void_p), ("nbr_final", C.c_void_p), ("pos_final", C.c_void_p)]


class DroneStepCall(C.Structure):
    """Mirror of `struct DroneStepCall` (include/dronesim.h): dronesim_step_ex's arguments, marshalled once."""
    _fields_ = [("p", C.c_void_p), ("ctl", C.c_void_p), ("pos", C.c_void_p), ("vel", C.c_void_p), ("t", C.c_void_p),
                ("reward", C.c_void_p), ("true_reward", C.c_void_p), ("z", C.c_void_p), ("nbr_idx", C.c_void_p),
                ("n_coll", C.c_void_p), ("done", C.c_void_p), ("E", C.c_int32), ("reserved", C.c_int32)]


class DroneParamsF64(C.Structure):
    """Mirror of `struct DroneParamsF64` (include/dronesim_verify.h): the float64 verification variant."""
    _fields_ = [("N", C.c_int32), ("k", C.c_int32), ("c", C.c_int32), ("max_steps", C.c_int32),
                ("dt", C.c_double), ("q", C.c_double), ("b", C.c_double), ("done_radius", C.c_double),
                ("ghost_factor", C.c_double),
                ("xF", C.c_void_p), ("d_hat", C.c_void_p), ("delta", C.c_void_p), ("radius", C.c_void_p)]


class DroneMlp(C.Structure):
    """Mirror of `struct DroneMlp` (include/dronesim.h)."""
    _fields_ = [("N", C.c_int32), ("d_in", C.c_int32), ("h1", C.c_int32), ("h2", C.c_int32), ("nout", C.c_int32),
                ("out_kind", C.c_int32), ("sample_kind", C.c_int32), ("w2_layout", C.c_int32),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("w3", C.c_void_p), ("b3", C.c_void_p)]


class DroneMlpBf16(C.Structure):
    """Mirror of `struct DroneMlpBf16` (include/dronesim.h)."""
    _fields_ = [("N", C.c_int32), ("d_in", C.c_int32), ("h1", C.c_int32), ("h2", C.c_int32), ("nout", C.c_int32),
                ("out_kind", C.c_int32), ("sample_kind", C.c_int32), ("reserved", C.c_int32),
                ("w1p", C.c_void_p), ("w2p", C.c_void_p), ("w3p", C.c_void_p),
                ("b1", C.c_void_p), ("b2", C.c_void_p), ("b3", C.c_void_p), ("wscale", C.c_void_p)]


class DroneSimError(RuntimeError):
    def __init__(self, code, where, detail):
        super().__init__(f"{where} failed with code {code}: {detail}")
        self.code = code


_lib = None


def lib():
    """Load libdronesim.so (built by __graft_entry__.build() / `make -C <pkg>/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "scalable_collision_avoidance_rl_amd/csrc`) with hipcc for gfx950. "
            "There is no CPU fallback for this path.")
    # torch first: the library's HIP calls must bind to the HIP runtime torch ships and initialises (the buffers and
    # streams handed across the ABI live there); loaded on its own, the library would pull in the system's copy and
    # see no context ("no ROCm-capable device is detected" at the first launch)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, u64, i64, f32 = C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_float
    P = C.POINTER(DroneParams)
    L.dronesim_step.argtypes = [P] + [vp] * 10 + [i32, vp]
    L.dronesim_observe.argtypes = [P] + [vp] * 8 + [i32, vp]
    L.dronesim_rollout.argtypes = [P] + [vp] * 10 + [i32, i32, vp]
    L.dronesim_control.argtypes = [P, i32, vp, vp, f32, i32, vp]
    L.dronesim_control.restype = C.c_int
    L.dronesim_episode_stats.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    L.dronesim_returns.argtypes = [vp, vp, f32, vp, i32, i32, i32, vp]
    L.dronesim_advantage.argtypes = [vp, vp, vp, vp, f32, vp, i32, i32, i32, i32, vp]
    L.dronesim_returns.restype = L.dronesim_advantage.restype = C.c_int
    L.dronesim_mlp_forward.argtypes = [C.POINTER(DroneMlp), vp, vp, vp, vp, u64, u64, i64, vp, vp, i32, vp]
    L.dronesim_mlp_forward.restype = C.c_int
    L.dronesim_mlp_forward_bf16.argtypes = [C.POINTER(DroneMlpBf16), vp, vp, vp, vp, u64, u64, i64, vp, vp, i32, vp]
    L.dronesim_mlp_forward_bf16.restype = C.c_int
    L.dronesim_mlp_forward_bf16x3.argtypes = L.dronesim_mlp_forward_bf16.argtypes
    L.dronesim_mlp_forward_bf16x3.restype = C.c_int
    L.dronesim_mlp_forward_f16x2.argtypes = L.dronesim_mlp_forward_bf16.argtypes
    L.dronesim_mlp_forward_f16x2.restype = C.c_int
    L.dronesim_mlp_bf16x3_stages.argtypes = [C.c_int, C.c_int]
    L.dronesim_mlp_rt_blocks.argtypes = [C.c_int, C.c_int, C.c_int]
    L.dronesim_mlp_rt16_blocks.argtypes = [C.c_int, C.c_int, C.c_int]
    L.dronesim_mlp_forward_f16x2_rt.argtypes = L.dronesim_mlp_forward_bf16.argtypes
    L.dronesim_mlp_forward_f16x2_rt.restype = C.c_int
    L.dronesim_mlp_bf16x3_stages.restype = C.c_int
    L.dronesim_reset.argtypes = [P, i32, i32, f32, u64, i64] + [vp] * 6 + [i32, vp]
    PC = C.POINTER(DroneEpisodeCtl)
    L.dronesim_step_ex.argtypes = [P, PC] + [vp] * 10 + [i32, vp]
    L.dronesim_step_call.argtypes = [vp, vp, vp]      # (plain integers: no per-call ctypes objects)
    L.dronesim_step_call.restype = C.c_int
    L.dronesim_rollout_ex.argtypes = [P, PC] + [vp] * 10 + [i32, i32, vp]
    L.dronesim_rollout_random.argtypes = [P, PC] + [vp] * 10 + [i32, i32, vp]
    L.dronesim_reset_ex.argtypes = [P, PC] + [vp] * 5 + [i32, vp]
    L.dronesim_reset_observe.argtypes = [P, PC] + [vp] * 7 + [i32, vp]
    L.dronesim_episode_reduce.argtypes = [vp, i32, vp, vp]
    for name in ("dronesim_step", "dronesim_observe", "dronesim_rollout", "dronesim_reset", "dronesim_step_ex",
                 "dronesim_rollout_ex", "dronesim_rollout_random", "dronesim_reset_ex", "dronesim_reset_observe", "dronesim_episode_reduce",
                 "dronesim_episode_stats", "dronesim_version"):
        getattr(L, name).restype = C.c_int
    L.dronesim_last_error.restype = C.c_char_p
    L.dronesim_error_string.restype = C.c_char_p
    L.dronesim_error_string.argtypes = [C.c_int]
    _lib = L
    return L


_vlib = None


def verify_lib():
    """Load libdronesim_verify.so (include/dronesim_verify.h): the float64 verification variant, test infrastructure."""
    global _vlib
    if _vlib is not None:
        return _vlib
    if not os.path.exists(VERIFY_LIB_PATH):
        raise ImportError(f"{VERIFY_LIB_PATH} not found: run `make -C scalable_collision_avoidance_rl_amd/csrc` "
                          "(hipcc, gfx950).  There is no CPU fallback.")
    import torch  # noqa: F401  (same HIP runtime as the product library, see lib())
    L = C.CDLL(VERIFY_LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    P64 = C.POINTER(DroneParamsF64)
    L.dronesim_step_f64.argtypes = [P64] + [vp] * 10 + [i32, vp]
    L.dronesim_observe_f64.argtypes = [P64] + [vp] * 7 + [i32, vp]
    L.dronesim_step_f64.restype = L.dronesim_observe_f64.restype = C.c_int
    L.dronesim_verify_last_error.restype = C.c_char_p
    _vlib = L
    return L


def check_verify(rc, where):
    if rc != OK:
        raise DroneSimError(rc, where, f"{lib().dronesim_error_string(rc).decode()} -- "
                                       f"{verify_lib().dronesim_verify_last_error().decode()}")


def check(rc, where):
    if rc != OK:
        L = lib()
        raise DroneSimError(rc, where, f"{L.dronesim_error_string(rc).decode()} -- "
                                       f"{L.dronesim_last_error().decode()}")
