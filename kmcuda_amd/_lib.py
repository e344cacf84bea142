"""ctypes loader for the HIP library (kmcuda_amd/libKMCUDA.so).

There is NO fallback: if the gfx950 library is missing or does not load, importing the step
API raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C kmcuda_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KMCUDA_AMD_LIB: another build of the same library (A/B runs of kernel variants, scripts/coarse_variants.sh)
LIB_PATH = os.environ.get("KMCUDA_AMD_LIB") or os.path.join(_HERE, "libKMCUDA.so")

u32, i32, f32 = ctypes.c_uint32, ctypes.c_int32, ctypes.c_float
vp = ctypes.c_void_p

_lib = None

# every symbol include/kmcuda.h and include/kmcuda_amd.h declare
EXPORTS = [
    "kmeans_cuda", "knn_cuda",
    "kmamd_engine_create", "kmamd_engine_destroy", "kmamd_engine_stream", "kmamd_engine_sync",
    "kmamd_lloyd_assign", "kmamd_lloyd_assign_exact", "kmamd_set_half_rows", "kmamd_set_row_cache", "kmamd_profile_read_coarse", "kmamd_set_filter", "kmamd_counters_read", "kmamd_counters_reset", "kmamd_yy_hint_stats",
    "kmamd_move_deltas", "kmamd_apply_delta", "kmamd_transpose", "kmamd_afkmc2_draws", "kmamd_reduce_len", "kmamd_reduce_fill",
    "kmamd_reduce_apply", "kmamd_reduce_apply_stop", "kmamd_reduce_apply_prepare", "kmamd_stop_report", "kmamd_stop_clear", "kmamd_centroids_written", "kmamd_set_carry", "kmamd_carry_stats", "kmamd_carry_pair_stats", "kmamd_duo_rows", "kmamd_carry_policy_sim", "kmamd_set_update_mode", "kmamd_last_run_stats", "kmamd_last_run_collective", "kmamd_adjust_exact", "kmamd_yy_configure", "kmamd_yy_init", "kmamd_yy_drifts", "kmamd_yy_filters",
    "kmamd_copy_to_device", "kmamd_profile_reset", "kmamd_profile_read", "kmamd_profile_enable", "kmamd_filter_kind", "kmamd_build_arch",
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "kmcuda_amd: %s is missing -- the HIP library has not been built "
            "(run `make -C kmcuda_amd/csrc`); there is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    L.kmeans_cuda.restype = i32
    L.kmeans_cuda.argtypes = [i32, vp, f32, f32, i32, u32, ctypes.c_uint16, u32, u32, u32, i32, i32, i32,
                              vp, vp, vp, vp]
    L.knn_cuda.restype = i32
    L.knn_cuda.argtypes = [ctypes.c_uint16, i32, u32, ctypes.c_uint16, u32, u32, i32, i32, i32, vp, vp, vp, vp]
    L.kmamd_engine_create.restype = i32
    L.kmamd_engine_create.argtypes = [ctypes.POINTER(vp), i32, u32, u32, u32, i32, i32, vp]
    L.kmamd_engine_destroy.restype = None
    L.kmamd_engine_destroy.argtypes = [vp]
    L.kmamd_engine_stream.restype = vp
    L.kmamd_engine_stream.argtypes = [vp]
    L.kmamd_engine_sync.restype = i32
    L.kmamd_engine_sync.argtypes = [vp]
    for name in ("kmamd_lloyd_assign", "kmamd_lloyd_assign_exact"):
        fn = getattr(L, name)
        fn.restype = i32
        fn.argtypes = [vp, vp, vp, vp, vp]
    L.kmamd_set_filter.restype = i32
    L.kmamd_set_filter.argtypes = [vp, i32]
    L.kmamd_set_half_rows.restype = i32
    L.kmamd_set_half_rows.argtypes = [vp, vp]
    L.kmamd_set_row_cache.restype = i32
    L.kmamd_set_row_cache.argtypes = [vp, i32]
    L.kmamd_counters_read.restype = i32
    L.kmamd_counters_read.argtypes = [vp, ctypes.POINTER(u32)]
    L.kmamd_counters_reset.restype = i32
    L.kmamd_counters_reset.argtypes = [vp, i32]
    L.kmamd_yy_hint_stats.restype = i32
    L.kmamd_yy_hint_stats.argtypes = [vp, vp]
    L.kmamd_move_deltas.restype = i32
    L.kmamd_move_deltas.argtypes = [vp, vp, vp, vp, vp, vp]
    L.kmamd_apply_delta.restype = i32
    L.kmamd_apply_delta.argtypes = [vp, vp, vp, vp, vp]
    L.kmamd_reduce_len.restype = ctypes.c_size_t
    L.kmamd_reduce_len.argtypes = [vp]
    L.kmamd_reduce_fill.restype = i32
    L.kmamd_reduce_fill.argtypes = [vp, vp, vp, vp, vp]
    L.kmamd_reduce_apply.restype = i32
    L.kmamd_reduce_apply.argtypes = [vp, vp, vp, vp]
    L.kmamd_reduce_apply_stop.restype = i32
    L.kmamd_reduce_apply_stop.argtypes = [vp, vp, vp, vp, f32, u32]
    L.kmamd_reduce_apply_prepare.restype = i32
    L.kmamd_reduce_apply_prepare.argtypes = [vp, vp, vp, vp, f32, u32]
    L.kmamd_stop_report.restype = i32
    L.kmamd_stop_report.argtypes = [vp, u32, ctypes.POINTER(u32)]
    L.kmamd_stop_clear.restype = i32
    L.kmamd_stop_clear.argtypes = [vp]
    L.kmamd_centroids_written.restype = i32
    L.kmamd_centroids_written.argtypes = [vp]
    L.kmamd_set_carry.restype = i32
    L.kmamd_set_carry.argtypes = [vp, i32]
    L.kmamd_carry_stats.restype = i32
    L.kmamd_carry_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
    L.kmamd_carry_pair_stats.restype = i32
    L.kmamd_carry_pair_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.kmamd_duo_rows.restype = i32
    L.kmamd_duo_rows.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    L.kmamd_carry_policy_sim.restype = i32
    L.kmamd_carry_policy_sim.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float, ctypes.POINTER(ctypes.c_uint32),
                                         ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32)]
    L.kmamd_set_update_mode.restype = i32
    L.kmamd_set_update_mode.argtypes = [vp, i32]
    L.kmamd_last_run_stats.restype = i32
    L.kmamd_last_run_stats.argtypes = [ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.kmamd_last_run_collective.restype = i32
    L.kmamd_last_run_collective.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u32)]
    L.kmamd_afkmc2_draws.restype = i32
    L.kmamd_afkmc2_draws.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, u32, u32, vp]
    L.kmamd_transpose.restype = i32
    L.kmamd_transpose.argtypes = [vp, vp, u32, u32, vp]
    L.kmamd_adjust_exact.restype = i32
    L.kmamd_adjust_exact.argtypes = [vp, vp, vp, vp, vp, vp]
    L.kmamd_yy_init.restype = i32
    L.kmamd_yy_configure.restype = i32
    L.kmamd_yy_configure.argtypes = [vp, u32, vp]
    L.kmamd_yy_init.argtypes = [vp, vp, vp, vp, vp]
    L.kmamd_yy_drifts.restype = i32
    L.kmamd_yy_drifts.argtypes = [vp, vp, vp, vp]
    L.kmamd_yy_filters.restype = i32
    L.kmamd_yy_filters.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.kmamd_profile_reset.restype = i32
    L.kmamd_profile_reset.argtypes = [vp]
    L.kmamd_filter_kind.restype = i32
    L.kmamd_filter_kind.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    L.kmamd_profile_enable.restype = i32
    L.kmamd_profile_enable.argtypes = [vp, i32]
    L.kmamd_profile_read_coarse.restype = i32
    L.kmamd_profile_read_coarse.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    L.kmamd_profile_read.restype = i32
    L.kmamd_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u32),
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.kmamd_copy_to_device.restype = i32
    L.kmamd_copy_to_device.argtypes = [i32, vp, vp, ctypes.c_size_t]
    L.kmamd_build_arch.restype = ctypes.c_char_p
    _lib = L
    return L


STATUS = {0: "Success", 1: "InvalidArguments", 2: "NoSuchDevice", 3: "MemoryAllocationFailure",
          4: "RuntimeError", 5: "MemoryCopyError"}


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, STATUS.get(rc, rc)))
