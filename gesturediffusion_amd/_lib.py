"""ctypes binding of libgdx.so (C ABI: include/gdx.h).

The product path has NO fallback: if the HIP library is missing or fails to load, every
compute entry point raises.  Build it with ``python __graft_entry__.py`` (or
``make -C gesturediffusion_amd/csrc``); the .so stays in-tree next to its sources.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GDX_LIBGDX: another build of the same library (A/B measurements of one kernel on one box); there is still no fallback
LIB_PATH = os.environ.get("GDX_LIBGDX") or os.path.join(_HERE, "csrc", "libgdx.so")

GDX_ARCH_MDM_OLD, GDX_ARCH_MDM = 1, 2
GDX_COND, GDX_UNCOND, GDX_CFG = 0, 1, 2
GDX_SAMPLER_P, GDX_SAMPLER_DDIM = 0, 1
GDX_ROW_PAD = 256      # include/gdx.h

EXPORTS = [
    "gdx_create", "gdx_destroy", "gdx_last_error", "gdx_set_weight", "gdx_weights_ready", "gdx_prepare",
    "gdx_set_condition", "gdx_forward", "gdx_set_keep_taps", "gdx_get_tap", "gdx_sampler_update", "gdx_q_sample",
    "gdx_randn", "gdx_sample_loop", "gdx_bench_ffn_gemm", "gdx_bench_gemm", "gdx_forward_flops", "gdx_profile_begin", "gdx_profile_end", "gdx_bench_attention",
    "gdx_linear_f16", "gdx_linear_f32", "gdx_bench_gemm_f16", "gdx_attention_f16", "gdx_attention_f32", "gdx_plms_update", "gdx_postprocess", "gdx_q_sample_t", "gdx_masked_l2", "gdx_set_graph_replay", "gdx_mfcc",
    "gdx_set_guards", "gdx_check_guards", "gdx_packed_bytes", "gdx_export_packed", "gdx_import_packed", "gdx_set_test_half_dtype",
    "gdx_set_test_gemmh_tile",
]


class GdxError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("arch", "njoints", "latent_dim", "ff_size", "num_layers", "num_heads",
                                         "seed_poses", "mfcc_dim", "cl_head", "window", "compute_dtype")]


class UpdateArgs(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("batch", C.c_int32), ("njoints", C.c_int32), ("frames", C.c_int32),
        ("coef", C.c_void_p), ("t", C.c_void_p), ("step_index", C.c_int32),
        ("x", C.c_void_p), ("x0_cond", C.c_void_p), ("x0_uncond", C.c_void_p), ("scale", C.c_void_p),
        ("inpaint_mask", C.c_void_p), ("inpaint_motion", C.c_void_p), ("noise", C.c_void_p),
        ("const_noise", C.c_int32), ("philox_seed", C.c_uint64), ("sample_offset", C.c_uint64),
        ("rng_step", C.c_uint32), ("out", C.c_void_p), ("pred_xstart", C.c_void_p),
        ("cond_grad", C.c_void_p), ("cond_coef", C.c_void_p), ("clip_denoised", C.c_int32),
    ]


class PlmsArgs(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("batch", C.c_int32), ("per_sample", C.c_int64), ("coef", C.c_void_p), ("t", C.c_void_p),
        ("step_index", C.c_int32), ("x", C.c_void_p), ("pred_xstart", C.c_void_p), ("eps", C.c_void_p * 4),
        ("out", C.c_void_p),
    ]


class LoopArgs(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("mode", C.c_int32), ("num_steps", C.c_int32), ("first_index", C.c_int32),
        ("coef", C.c_void_p), ("timestep_map", C.c_void_p), ("x", C.c_void_p), ("scale", C.c_void_p),
        ("inpaint_mask", C.c_void_p), ("inpaint_motion", C.c_void_p), ("noise_tape", C.c_void_p),
        ("const_noise", C.c_int32), ("philox_seed", C.c_uint64), ("sample_offset", C.c_uint64),
        ("dump", C.c_void_p), ("dump_steps", C.c_void_p), ("n_dump", C.c_int32),
        ("run_steps", C.c_int32), ("k_base", C.c_int32), ("clip_denoised", C.c_int32),
    ]


_lib = None


def load():
    """Load libgdx.so once; raise GdxError (never fall back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GdxError(f"HIP library not built: {LIB_PATH} is missing "
                       f"(run `python __graft_entry__.py` or `make -C gesturediffusion_amd/csrc`)")
    try:
        # torch ships its own HIP runtime; importing it first makes libgdx.so bind to that same
        # runtime instance, so torch's streams / device pointers are valid inside the library.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64, wrong arch ...
        raise GdxError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, i64, u64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32
    lib.gdx_last_error.restype = C.c_char_p
    sigs = {
        "gdx_create": [C.POINTER(Config), C.POINTER(vp)],
        "gdx_destroy": [vp],
        "gdx_set_weight": [vp, C.c_char_p, vp, C.POINTER(i64), i32, vp],
        "gdx_weights_ready": [vp],
        "gdx_prepare": [vp, i32, i32],
        "gdx_set_condition": [vp, vp, vp, vp],
        "gdx_forward": [vp, vp, vp, i32, vp, vp, vp],
        "gdx_set_keep_taps": [vp, i32],
        "gdx_get_tap": [vp, i32, vp, i64, vp],
        "gdx_sampler_update": [C.POINTER(UpdateArgs), vp],
        "gdx_q_sample": [vp, vp, vp, i32, i64, vp, vp],
        "gdx_randn": [vp, i32, i64, u64, u64, u32, vp],
        "gdx_sample_loop": [vp, C.POINTER(LoopArgs), vp],
        "gdx_bench_ffn_gemm": [vp, i32, C.POINTER(C.c_float), vp],
        "gdx_forward_flops": [vp, i32, C.POINTER(C.c_double)],
        "gdx_bench_gemm": [i32, i32, i32, i32, i32, C.POINTER(C.c_float), vp],
        "gdx_bench_attention": [i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), vp],
        "gdx_linear_f16": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "gdx_linear_f32": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
        "gdx_set_graph_replay": [vp, i32],
        "gdx_mfcc": [vp, i64, i32, i32, i32, i32, i32, i32, C.c_float, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "gdx_q_sample_t": [vp, vp, vp, vp, i32, i64, vp, vp],
        "gdx_masked_l2": [vp, vp, vp, vp, i32, i32, i32, vp],
        "gdx_postprocess": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
        "gdx_plms_update": [C.POINTER(PlmsArgs), vp],
        "gdx_attention_f16": [vp, vp, i32, i32, i32, i32, vp],
        "gdx_attention_f32": [vp, vp, i32, i32, i32, i32, i32, vp],
        "gdx_bench_gemm_f16": [i32, i32, i32, i32, i32, C.POINTER(C.c_float), vp],
        "gdx_set_guards": [vp, i32],
        "gdx_check_guards": [vp, C.POINTER(i64), C.POINTER(i32), vp],
        "gdx_set_test_half_dtype": [i32],
        "gdx_set_test_gemmh_tile": [i32, i32],
        "gdx_packed_bytes": [vp, C.POINTER(i64)],
        "gdx_export_packed": [vp, vp, i64, vp],
        "gdx_import_packed": [vp, vp, i64, vp],
        "gdx_profile_begin": [vp, i32],
        "gdx_profile_end": [vp, C.POINTER(C.c_float), C.POINTER(i32)],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load()
        raise GdxError(lib.gdx_last_error().decode() or f"libgdx error {rc}")
