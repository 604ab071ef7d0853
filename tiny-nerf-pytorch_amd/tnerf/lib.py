"""ctypes binding of libtnerf_hip.so (include/tnerf.h).  There is NO fallback: if the library is
missing or a call fails, a RuntimeError is raised."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TNERF_LIB") or os.path.join(_HERE, "libtnerf_hip.so")   # TNERF_LIB: A/B a diagnostic build

OK, EINVAL, EUNSUPPORTED, ESMALL = 0, -1, -2, -3


class MlpDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("hidden", C.c_int32), ("depth", C.c_int32), ("skip_at", C.c_int32), ("flags", C.c_int32)]


FLAG_FP32_MFMA = 1        # fp32 products on v_mfma_f32_32x32x2_f32 instead of the x3 scheme (three fp16 partial products; include/tnerf.h)


class Camera(C.Structure):
    _fields_ = [("c2w", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("focal", C.c_float),
                ("pix_index", C.c_void_p), ("pix_first", C.c_int64)]


class PlanSizes(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_params", "packed_floats", "stash_floats", "slab_floats", "job_ints",
                                         "reduce_ints", "n_jobs", "stash_row_stride")]


class Bf16Sizes(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("packed_bytes", "pack_entries", "n_fragments", "bias_offset_bytes", "n_fwd_fragments")]


class Bf16TrainPlan(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_tiles", "stash_bytes", "slab_floats", "job_ints", "reduce_ints", "n_jobs")]


class StepArgs(C.Structure):
    """tnerf_step_args (include/tnerf.h): the whole train step on device-resident state."""
    _fields_ = [("desc", MlpDesc), ("precision", C.c_int32), ("phases", C.c_int32),
                ("poses", C.c_void_p), ("pixels", C.c_void_p), ("n_images", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("focal", C.c_float),
                ("n_rays", C.c_int64), ("ray_first", C.c_int64), ("n_rays_global", C.c_int64), ("n_samples", C.c_int32), ("white_bkgd", C.c_int32),
                ("ztab", C.c_void_p), ("seed", C.c_uint64), ("loss_denominator", C.c_double), ("step", C.c_void_p),
                ("packed", C.c_void_p),
                ("comp_rgb", C.c_void_p), ("ray_ws", C.c_void_p), ("ray_ws_floats", C.c_int64), ("pix_out", C.c_void_p), ("loss_out", C.c_void_p),
                ("stash", C.c_void_p), ("stash_row_stride", C.c_int64), ("stash_capacity", C.c_int64),
                ("job_table", C.c_void_p), ("n_jobs", C.c_int64), ("slabs", C.c_void_p), ("reduce_table", C.c_void_p), ("grads", C.c_void_p),
                ("params", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("scatter_table", C.c_void_p), ("scatter_width", C.c_int32),
                ("packed_x3", C.c_void_p), ("scatter_x3", C.c_void_p), ("scatter_x3_width", C.c_int32),
                ("pack_x3", C.c_void_p)]


PHASE_GRADIENT, PHASE_REDUCE, PHASE_UPDATE = 1, 2, 4
ABI_VERSION = 3

_P = C.c_void_p           # device pointers travel as integers (tensor.data_ptr())
_I32, _I64, _U64, _F, _D = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
_DESC = C.POINTER(MlpDesc)

# name -> (restype, argtypes); mirrors include/tnerf.h one to one
SIGNATURES = {
    "tnerf_version": (C.c_int, []),
    "tnerf_last_error_string": (C.c_char_p, []),
    "tnerf_sample_tables": (C.c_int, [_F, _F, _I32, _P, _P]),
    "tnerf_param_count": (_I64, [_DESC]),
    "tnerf_param_layout": (C.c_int, [_DESC, _P, _P, _P]),
    "tnerf_input_pairing": (C.c_int, [_I32, _P, _P]),
    "tnerf_plan_sizes_query": (C.c_int, [_DESC, _I64, _I32, C.POINTER(PlanSizes)]),
    "tnerf_plan_fill": (C.c_int, [_DESC, _I64, _I32, _P, _P, _P]),
    "tnerf_get_rays": (C.c_int, [_I32, _I32, _F, _P, _P, _P, _P]),
    "tnerf_sample_encode_fwd": (C.c_int, [_P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _P, _P, _P, _I32, _I32, _P]),
    "tnerf_sample_per_ray_fwd": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _I32, _P, _U64, _U64, _P, _P, _P]),
    "tnerf_posenc_fwd": (C.c_int, [_P, _I64, _I32, _I32, _P, _P]),
    "tnerf_composite_fwd": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P]),
    "tnerf_composite_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "tnerf_mlp_pack": (C.c_int, [_P, _P, _I64, _P, _P]),
    "tnerf_mlp_fwd": (C.c_int, [_DESC, _P, _P, _I64, _P, _P, _P, _I64, _P]),
    "tnerf_mlp_bwd": (C.c_int, [_DESC, _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P]),
    "tnerf_render_fused": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _P, _P]),
    "tnerf_train_fwd_fused": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _I64, _P]),
    "tnerf_train_bwd_fused": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _I64,
                                        _P, _I64, _P, _P, _P, _P, _P]),
    "tnerf_train_dgrad_fused": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _I64, _P]),
    "tnerf_wgrad": (C.c_int, [_DESC, _P, _I64, _I64, _P, _I64, _P, _P]),
    "tnerf_wgrad_reduce": (C.c_int, [_P, _P, _I64, _P, _P]),
    "tnerf_train_ws_floats": (C.c_int64, [_I64]),
    "tnerf_train_step_fused": (C.c_int, [_DESC, _P, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _D,
                                         _P, _P, _I64, _P, _P, _I64, _P, _I64, _P, _P, _P, _P, _P]),
    "tnerf_render_fused_cam": (C.c_int, [_DESC, _P, C.POINTER(Camera), _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _P, _P]),
    "tnerf_train_step_fused_cam": (C.c_int, [_DESC, _P, C.POINTER(Camera), _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _D,
                                             _P, _P, _I64, _P, _P, _I64, _P, _I64, _P, _P, _P, _P, _P]),
    "tnerf_bf16_plan_sizes": (C.c_int, [_DESC, C.POINTER(Bf16Sizes)]),
    "tnerf_bf16_pack_table": (C.c_int, [_DESC, _P]),
    "tnerf_mlp_pack_bf16": (C.c_int, [_DESC, _P, _P, _P, _P]),
    "tnerf_render_fused_bf16": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _P, _P]),
    "tnerf_render_fused_cam_bf16": (C.c_int, [_DESC, _P, C.POINTER(Camera), _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _P, _P]),
    "tnerf_bf16_train_sizes": (C.c_int, [_DESC, _I64, _I32, _I32, C.POINTER(Bf16TrainPlan)]),
    "tnerf_bf16_train_fill": (C.c_int, [_DESC, _I64, _I32, _I32, _P, _P]),
    "tnerf_train_fwd_fused_bf16": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _P]),
    "tnerf_train_dgrad_fused_bf16": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _P]),
    "tnerf_wgrad_bf16": (C.c_int, [_DESC, _P, _I64, _P, _I64, _P, _P]),
    "tnerf_train_step_fused_bf16": (C.c_int, [_DESC, _P, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _D,
                                              _P, _P, _I64, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "tnerf_train_step_fused_cam_bf16": (C.c_int, [_DESC, _P, C.POINTER(Camera), _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _D,
                                                  _P, _P, _I64, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "tnerf_composite_bwd_geom": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "tnerf_sample_bwd": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _P, _P, _P, _P]),
    "tnerf_posenc_bwd": (C.c_int, [_P, _I64, _I32, _I32, _P, _P, _P]),
    "tnerf_get_rays_bwd_scratch_floats": (C.c_int64, [_I32, _I32]),
    "tnerf_get_rays_bwd": (C.c_int, [_I32, _I32, _F, _P, _P, _P, _I64, _P, _P]),
    "tnerf_mlp_generic_acts_floats": (C.c_int64, [_DESC, _I64]),
    "tnerf_mlp_generic_scratch_floats": (C.c_int64, [_DESC, _I64]),
    "tnerf_mlp_fwd_generic": (C.c_int, [_DESC, _P, _P, _P, _I64, _P, _P, _P, _I64, _P]),
    "tnerf_mlp_bwd_generic": (C.c_int, [_DESC, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _P]),
    "tnerf_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I64, _F, _P]),
    "tnerf_x3_plan_sizes": (C.c_int, [_DESC, C.POINTER(Bf16Sizes)]),
    "tnerf_x3_pack_table": (C.c_int, [_DESC, _P]),
    "tnerf_mlp_pack_x3": (C.c_int, [_DESC, _P, _P, _P, _P]),
    "tnerf_mlp_pack_x3_floor": (C.c_int, [_DESC, _P, _P, _P, C.c_float, _P]),
    "tnerf_x3_domain_counts": (C.c_int, [_DESC, _P, _P, _P, _P, _P]),
    "tnerf_render_fused_x3": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _P, _P]),
    "tnerf_render_fused_cam_x3": (C.c_int, [_DESC, _P, C.POINTER(Camera), _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _P, _P]),
    "tnerf_train_fwd_fused_x3": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _I64, _P]),
    "tnerf_mlp_fwd_x3": (C.c_int, [_DESC, _P, _P, _I64, _P, _P, _P, _I64, _P]),
    "tnerf_mlp_bwd_x3": (C.c_int, [_DESC, _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P]),
    "tnerf_train_dgrad_fused_x3": (C.c_int, [_DESC, _P, _P, _P, _I64, _I32, _P, _I32, _P, _U64, _U64, _I32, _P, _P, _I64, _P]),
    "tnerf_train_step_dataset": (C.c_int, [C.POINTER(StepArgs), _P]),
    "tnerf_graph_begin": (C.c_int, [_P]),
    "tnerf_graph_end": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "tnerf_graph_launch": (C.c_int, [_P, _P]),
    "tnerf_graph_destroy": (C.c_int, [_P]),
    "tnerf_comm_unique_id": (C.c_int, [_P]),
    "tnerf_comm_init_rank": (C.c_int, [_P, _I32, _I32, C.POINTER(C.c_void_p)]),
    "tnerf_comm_destroy": (C.c_int, [_P]),
    "tnerf_allreduce_grads": (C.c_int, [_P, _P, _I64, _P]),
}

_lib = None


def load():
    """Load libtnerf_hip.so once; raise (never fall back) if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                "(tiny-nerf-pytorch_amd/csrc/build.sh).  The HIP path has no CPU fallback.")
        host_only = bool(os.environ.get("TNERF_HOST_ONLY"))      # sanitizer build of csrc/host_plan.cpp alone (tests/test_host_sanitizers.py)
        if not host_only:
            # torch FIRST: a ROCm torch wheel ships its own libamdhip64 under torch/lib.  If this library is loaded before torch is
            # imported, it pulls in the system's copy, torch then brings its own, and the process holds two HIP runtimes — the second one
            # to initialise finds "no ROCm-capable device" (seen when build() and smoke() ran in one process).  With torch's copy already
            # mapped, this library's dependency resolves to it.
            import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)      # AttributeError if the .so does not export it
            except AttributeError:
                if host_only:
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        if lib.tnerf_version() != ABI_VERSION:
            raise RuntimeError(f"libtnerf_hip.so ABI version {lib.tnerf_version()} != {ABI_VERSION}: rebuild it (python __graft_entry__.py build)")
        _lib = lib
    return _lib


def last_error() -> str:
    return load().tnerf_last_error_string().decode()


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    msg = last_error()
    if rc == EUNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    if rc < 0:
        raise ValueError(f"{what}: {msg} (code {rc})")
    raise RuntimeError(f"{what}: HIP/RCCL error {rc}: {msg}")


def train_ws_floats(n_rays: int) -> int:
    """tnerf_train_ws_floats: floats of the per-ray workspace (g_comp_ws / tnerf_step_args.ray_ws) for n_rays rays."""
    n = int(load().tnerf_train_ws_floats(int(n_rays)))
    if n < 0:
        raise TnerfError(n, "tnerf_train_ws_floats", last_error())
    return n


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
