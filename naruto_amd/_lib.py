"""ctypes binding of libnaruto_hip.so (C ABI: include/naruto_hip.h).

This is the ONLY compute backend of the package: if the shared library has not been built the import
fails loudly -- there is no PyTorch / CPU fallback for the hot path.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NARUTO_HIP_LIB") or os.path.join(_HERE, "libnaruto_hip.so")      # override: kernel experiments only
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["naruto_api.hip", "naruto_field.hip", "naruto_binned.hip", "naruto_render.hip", "naruto_rays.hip", "naruto_train.hip", "naruto_renderfused.hip", "naruto_planner.hip", "naruto_mesh.hip", "naruto_parts.hip",
           "naruto_mc_table.inc", "naruto_common.h"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "naruto_hip.h")

MAX_LEVELS = 16
LOSS_NSUMS = 16
LOSS_SLOT_MINUNCERT = 9


class NarutoFieldDesc(C.Structure):
    _fields_ = [
        ("n_levels", C.c_uint32), ("n_features", C.c_uint32), ("log2_hashmap_size", C.c_uint32),
        ("base_resolution", C.c_uint32), ("per_level_scale", C.c_float), ("n_bins", C.c_uint32),
        ("hidden_dim", C.c_uint32), ("geo_feat_dim", C.c_uint32), ("hidden_dim_color", C.c_uint32),
        ("uncert_dims", C.c_uint32 * 3), ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
        ("trunc", C.c_float), ("sc_factor", C.c_float), ("white_bkgd", C.c_int32), ("mlp_mode", C.c_uint32),
    ]


class NarutoParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("table", "uncert_grid", "sdf_w0", "sdf_w1", "col_w0", "col_w1")]


class NarutoGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("table", "uncert_grid", "sdf_w0", "sdf_w1", "col_w0", "col_w1")]


class NarutoExtraPoints(C.Structure):
    _fields_ = [("x", C.c_void_p), ("d_feat", C.c_void_p), ("scale", C.c_void_p), ("n", C.c_uint32)]


class NarutoAdamSeg(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_uint64), ("lr", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("step_lag", C.c_uint32)]


MLP_FP32, MLP_BF16 = 0, 1
BWD_OVERWRITE_WEIGHT_GRADS = 1
BWD_OVERWRITE_TABLE_GRAD = 2
TRAIN_BWD_MLP_ONLY, TRAIN_BWD_TABLE_ONLY = 4, 8
TRAIN_FWD_DEFER_TAIL, TRAIN_BWD_DEFERRED_TAIL, TRAIN_BWD_SUMS_GIVEN = 2, 16, 32
TRAIN_FWD_SUMS_TV_LATER, TRAIN_BWD_TV_MOVED = 3, 64
ADAM_ADVANCE = 1
ADAM_ZERO_GRAD = 2


class NarutoRayBatch(C.Structure):
    _fields_ = [("store", C.c_void_p), ("n_kf", C.c_uint32), ("rays_per_kf", C.c_uint32), ("frame_ids", C.c_void_p),
                ("keyframe_every", C.c_int64), ("n_global", C.c_uint32), ("current", C.c_void_p), ("cur_list", C.c_void_p),
                ("n_cur_pop", C.c_uint64), ("n_cur", C.c_uint32), ("poses", C.c_void_p), ("n_poses", C.c_uint32),
                ("seed", C.c_uint64), ("counter", C.c_uint64), ("rays_o", C.c_void_p), ("rays_d", C.c_void_p),
                ("target_s", C.c_void_p), ("target_d", C.c_void_p), ("ids_out", C.c_void_p), ("rng", C.c_void_p), ("dyn", C.c_void_p),
                ("keys_out", C.c_void_p), ("key_base", C.c_uint32), ("key_tail", C.c_uint32), ("key_vol", C.c_void_p), ("key_dims", C.c_uint32 * 3),
                ("key_bbox_min", C.c_float * 3), ("key_voxel_scale", C.c_float)]


class NarutoFusedAdam(C.Structure):
    _fields_ = [("param", C.c_void_p * 5), ("exp_avg", C.c_void_p * 5), ("exp_avg_sq", C.c_void_p * 5),
                ("lr", C.c_float * 5), ("eps", C.c_float * 5), ("weight_decay", C.c_float * 5),
                ("beta1", C.c_float), ("beta2", C.c_float), ("step_dev", C.c_void_p), ("next_batch", C.c_void_p)]


class NarutoTrainStep(C.Structure):
    _fields_ = [
        ("n_rays", C.c_uint32), ("n_samples_d", C.c_uint32), ("n_range_d", C.c_uint32), ("perturb", C.c_uint32),
        ("near_", C.c_float), ("far_", C.c_float), ("range_d", C.c_float), ("depth_trunc", C.c_float), ("rgb_missing", C.c_float),
        ("smooth_points", C.c_uint32), ("smooth_voxel", C.c_float), ("smooth_margin", C.c_float), ("smooth_grad_scale", C.c_float),
        ("n_rays_total", C.c_uint64),
        ("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("target_rgb", C.c_void_p), ("target_d", C.c_void_p),
        ("rand", C.c_void_p), ("rand6", C.c_void_p), ("rng", C.c_void_p), ("loss_weights", C.c_void_p),
        ("z_vals", C.c_void_p), ("raw", C.c_void_p), ("feat_save", C.c_void_p),
        ("rgb", C.c_void_p), ("depth", C.c_void_p), ("uncert_map", C.c_void_p),
        ("sums", C.c_void_p), ("losses", C.c_void_p), ("d_raw", C.c_void_p),
        ("ray_count", C.c_void_p), ("ray_offset", C.c_void_p), ("active_idx", C.c_void_p), ("n_active", C.c_void_p),
        ("workspace", C.c_void_p), ("loss_weight_parts", C.c_void_p * 10), ("min_uncert_running", C.c_void_p),
    ]


class NarutoRender(C.Structure):
    _fields_ = [("n_rays", C.c_uint32), ("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("target_d", C.c_void_p),
                ("near_", C.c_float), ("far_", C.c_float), ("n_samples_d", C.c_uint32), ("n_range_d", C.c_uint32), ("range_d", C.c_float),
                ("n_samples", C.c_uint32), ("rand", C.c_void_p), ("rng", C.c_void_p),
                ("rgb", C.c_void_p), ("depth", C.c_void_p), ("disp", C.c_void_p), ("acc", C.c_void_p), ("depth_var", C.c_void_p),
                ("uncert_map", C.c_void_p), ("weights", C.c_void_p), ("raw", C.c_void_p), ("z_vals", C.c_void_p)]


class NarutoPoints(C.Structure):
    _fields_ = [("x", C.c_void_p), ("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("z_vals", C.c_void_p),
                ("n_samples", C.c_uint32)]


def build_command(out: str = LIB_PATH) -> List[str]:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
            os.path.join(CSRC, "naruto_api.hip"), "-o", out]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or needs_build():
        tmp = f"{LIB_PATH}.{os.getpid()}.tmp"               # never a half-written library under the final name
        cmd = build_command(tmp)
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.run(cmd, check=True)
            os.replace(tmp, LIB_PATH)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB_PATH


_V, _U32, _U64, _F, _I = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int

# name -> (restype, argtypes); every symbol include/naruto_hip.h declares
SIGNATURES = {
    "naruto_last_error": (C.c_char_p, []),
    "naruto_version": (_I, []),
    "naruto_field_create": (_I, [C.POINTER(NarutoFieldDesc), C.POINTER(_V)]),
    "naruto_field_destroy": (None, [_V]),
    "naruto_field_levels": (_I, [_V, C.POINTER(_F), C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_U32)]),
    "naruto_field_n_entries": (_U64, [_V]),
    "naruto_sample_z": (_I, [_U32, _V, _F, _F, _U32, _U32, _F, _U32, _V, _V, _V]),
    "naruto_hash_encode_fwd": (_I, [_V, _U32, _V, _V, _V, _V]),
    "naruto_scatter_workspace": (C.c_size_t, [_V, _U32]),
    "naruto_field_scatter_overwrites": (C.c_int, [_V]),
    "naruto_hash_encode_bwd": (_I, [_V, _U32, _V, _V, _V, _V, _V, _V]),
    "naruto_smoothness_workspace": (C.c_size_t, [_U32]),
    "naruto_smoothness_fwd": (_I, [_V, _V, _U32, _F, _F, _V, _V, _V, _V, _V, _V]),
    "naruto_query_fwd": (_I, [_V, C.POINTER(NarutoParams), _U32, C.POINTER(NarutoPoints), _V, _V, _V, _V, _V]),
    "naruto_query_bwd_workspace": (C.c_size_t, [_V, _U32]),
    "naruto_query_bwd": (_I, [_V, C.POINTER(NarutoParams), _U32, C.POINTER(NarutoPoints), _V, _V, _V, _V, _V,
                              C.POINTER(NarutoExtraPoints), _U32, C.POINTER(NarutoGrads), _V, _V]),
    "naruto_active_ray_workspace": (C.c_size_t, [_U32, _U32]),
    "naruto_active_ray_select_keyed": (_I, [_U32, _U32, _U32, _U32, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V]),
    "naruto_active_ray_select": (_I, [_U32, _U32, _U32, _U32, _V, _V, _V, _V, _V, C.POINTER(_U32), C.POINTER(_F), _F, _V, _V, _V, _V, _V, _V]),
    "naruto_rays_to_world": (_I, [_U32, _V, _V, _V, _V, _V, _V]),
    "naruto_map_volumes": (_I, [_U32, _V, _V, _V]),
    "naruto_assemble_rays": (_I, [C.POINTER(NarutoRayBatch), _V]),
    "naruto_assemble_select": (_I, [C.POINTER(NarutoRayBatch), _U32, _U32, _U32, _V, C.POINTER(_U32), C.POINTER(_F), _F, _V, _V, _V, _V, _V]),
    "naruto_sample_distinct": (_I, [_U64, _U32, _U64, _U64, _V, _V]),
    "naruto_perm_index": (_U64, [_U64, _U64, _U64, _U64, _U64]),
    "naruto_goal_targets_workspace": (C.c_size_t, [_U32, _U32]),
    "naruto_goal_targets": (_I, [C.POINTER(_U32), _V, _U32, _U32, _V, _V, _V]),
    "naruto_goal_aggregate": (_I, [C.POINTER(_U32), _V, _V, _U32, _V, _U32, _V, _F, _F, _F, _V, _V, _V]),
    "naruto_lattice_points": (_I, [C.POINTER(_U32), _V, _V, _V, _V, _V]),
    "naruto_mesh_workspace": (C.c_size_t, [C.POINTER(_U32)]),
    "naruto_mesh_count": (_I, [C.POINTER(_U32), _V, C.c_double, C.c_double, _V, _V, _V]),
    "naruto_mesh_emit": (_I, [C.POINTER(_U32), _V, C.c_double, _V, _U64, _U64, _V, _V, _V]),
    "naruto_adam_multi": (_I, [C.POINTER(NarutoAdamSeg), _U32, _F, _F, _U32, _V, _U32, _V]),
    "naruto_train_workspace": (C.c_size_t, [_V, C.POINTER(NarutoTrainStep)]),
    "naruto_train_forward": (_I, [_V, C.POINTER(NarutoParams), C.POINTER(NarutoTrainStep), _I, _V]),
    "naruto_train_finalize": (_I, [_V, C.POINTER(NarutoTrainStep), _V]),
    "naruto_oneblob_fwd": (_I, [_V, C.c_uint32, _V, _V, _V]),
    "naruto_uncert_sample": (_I, [_V, C.c_uint32, _V, _V, _V, _V]),
    "naruto_debug_random_lines": (_I, [_V, C.c_uint64, C.c_uint32, _V, C.POINTER(C.c_uint64), _V]),
    "naruto_decoder_fwd": (_I, [_V, C.POINTER(NarutoParams), C.c_uint32, C.c_int, _V, _V, _V, _V]),
    "naruto_debug_train_query_fwd": (_I, [_V, C.POINTER(NarutoParams), C.POINTER(NarutoTrainStep), _V]),
    "naruto_debug_fwd_timeline": (_I, [_V]),
    "naruto_debug_train_scatter": (_I, [_V, C.POINTER(NarutoParams), C.POINTER(NarutoTrainStep), _V]),
    "naruto_render_fwd": (_I, [_V, C.POINTER(NarutoParams), C.POINTER(NarutoRender), _V]),
    "naruto_train_backward": (_I, [_V, C.POINTER(NarutoParams), C.POINTER(NarutoTrainStep), C.POINTER(NarutoGrads), _U32,
                                   C.POINTER(NarutoFusedAdam), _V]),
    "naruto_compact_active": (_I, [_U32, _U32, _V, _V, _V, _V, _V]),
    "naruto_composite_fwd": (_I, [_V, _U32, _U32, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V]),
    "naruto_composite_bwd": (_I, [_V, _U32, _U32, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _I, _V]),
    "naruto_loss_workspace": (C.c_size_t, [_U32]),
    "naruto_loss_sums": (_I, [_V, _U32, _U32, _V, _V, _V, _V, _V, _V, _V, _F, _F, _V, _V, _V, _V]),
    "naruto_loss_finalize": (_I, [_V, _U64, _U32, _V, _V]),
    "naruto_loss_bwd": (_I, [_V, _U32, _U32, _V, _V, _V, _V, _F, _F, _V, _U64, _V, _V, _V, _V]),
    "naruto_adam_step": (_I, [_V, _V, _V, _V, _U64, _F, _F, _F, _F, _F, _U32, _V, _V]),
    "naruto_debug_mfma_layout": (_I, [_V, _V, _V, _V]),
    "naruto_debug_permlane_swap": (_I, [_V, _V, _V, _V]),
    "naruto_debug_mfma_bf16_layout": (_I, [_V, _V, _V, _V]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no fallback path.")
        # The device pointers and streams handed to the library are PyTorch's, so both must talk to ONE HIP runtime: the
        # one PyTorch-ROCm brings along.  Loaded first, libnaruto_hip.so would pull in /opt/rocm's libamdhip64 and the
        # process would end up with two runtimes (the second one reports "no ROCm-capable device").  Import torch first.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class NarutoError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().naruto_last_error().decode("utf-8", "replace")
        raise NarutoError(f"{what or 'libnaruto_hip'} failed ({rc}): {msg}")


def kernel_resources(lib_path: str = None):
    """Per kernel of the device code object embedded in the built library: the AMDGPU metadata notes the loader reads
    (``.vgpr_count``, ``.agpr_count``, ``.vgpr_spill_count``, ``.sgpr_spill_count``, ``.private_segment_fixed_size`` = scratch bytes per
    lane, ``.group_segment_fixed_size`` = static LDS).  Uses the ROCm LLVM tools (llvm-objcopy, clang-offload-bundler, llvm-readelf);
    returns {demangled-ish kernel name: {field: int}}.  tests/test_host.py holds the hot kernels to zero scratch with it."""
    import re
    import tempfile
    llvm = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    lib_path = lib_path or LIB_PATH
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)\s*$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "agpr_count":                      # first field of a kernel's record (fields are sorted by name)
            cur = {"agpr_count": int(val)}
        elif cur is not None and key == "name":
            name = re.sub(r"^_ZN6naruto\d+", "", val)
            m2 = re.match(r"(\w+?)(?:I((?:L[bi]\d+E)+)E)?E", name)          # Itanium: name [I <Lb0E | Li128E ...> E] E <signature>
            if m2:
                name = m2.group(1)
                if m2.group(2):
                    args = re.findall(r"L([bi])(\d+)E", m2.group(2))
                    name += "<" + ",".join(("true" if v == "1" else "false") if k == "b" else v for k, v in args) + ">"
            out[name] = cur
        elif cur is not None and key in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "sgpr_count", "private_segment_fixed_size",
                                         "group_segment_fixed_size"):
            cur[key] = int(val)
    return out
