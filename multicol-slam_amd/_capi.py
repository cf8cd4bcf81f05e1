"""ctypes binding of libmcs_hip.so (the C ABI declared in include/mcs_c.h).

The library is the product: if it is missing or no HIP device is usable, everything here raises — there is no
CPU fallback (the CPU oracle under oracle/ is test infrastructure and is never imported from this package).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCS_HIP_LIB", os.path.join(_HERE, "libmcs_hip.so"))   # override only for A/B kernel experiments

MCS_OK, MCS_ERR_INVALID, MCS_ERR_HIP, MCS_ERR_CAPACITY, MCS_ERR_UNSUPPORTED = 0, -1, -2, -3, -4
MEM_HOST, MEM_DEVICE = 0, 1
MASKS_RESIDENT = C.c_void_p(1)   # MCS_MASKS_RESIDENT of include/mcs_c.h
MAX_POLY = 16


class McsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmcs_hip error %d: %s" % (code, msg))
        self.code = code


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float), ("response", C.c_float),
                ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


class Ocam(C.Structure):
    _fields_ = [("c", C.c_double), ("d", C.c_double), ("e", C.c_double), ("u0", C.c_double), ("v0", C.c_double),
                ("p", C.c_double * MAX_POLY), ("p_deg", C.c_int32), ("invP", C.c_double * MAX_POLY), ("invP_deg", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32)]


class ExtractorParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scaleFactor", C.c_float), ("nlevels", C.c_int32), ("edgeThreshold", C.c_int32),
                ("firstLevel", C.c_int32), ("scoreType", C.c_int32), ("patchSize", C.c_int32), ("fastThreshold", C.c_int32),
                ("useAgast", C.c_int32), ("fastAgastType", C.c_int32), ("do_dBrief", C.c_int32), ("learnMasks", C.c_int32),
                ("descSize", C.c_int32)]


class DescSet(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("mask", C.c_void_p), ("valid", C.c_void_p), ("group", C.c_void_p), ("n", C.c_int32),
                ("stride", C.c_int32), ("block_rows", C.c_int32), ("block_pitch_rows", C.c_int64)]


def make_ocam(cam):
    o = Ocam()
    o.c, o.d, o.e, o.u0, o.v0 = cam["c"], cam["d"], cam["e"], cam["u0"], cam["v0"]
    for i, v in enumerate(cam["p"]):
        o.p[i] = v
    o.p_deg = len(cam["p"])
    for i, v in enumerate(cam["invP"]):
        o.invP[i] = v
    o.invP_deg = len(cam["invP"])
    o.width, o.height = cam["width"], cam["height"]
    return o


EXPORTS = [
    "mcs_last_error", "mcs_abi_version", "mcs_device_count", "mcs_ctx_create", "mcs_ctx_destroy", "mcs_ctx_synchronize", "mcs_extractor_create",
    "mcs_extractor_destroy", "mcs_extractor_set_masks", "mcs_extractor_kp_capacity", "mcs_extractor_levels", "mcs_extract_batch", "mcs_extractor_status",
    "mcs_extractor_tap_level", "mcs_extractor_tap_candidates", "mcs_extractor_tap_selected", "mcs_match_topk",
    "mcs_match_topk_batched", "mcs_descriptor_distance", "mcs_descriptor_distance_masked", "mcs_ctx_enable_timing",
    "mcs_ctx_kernel_ms", "mcs_ctx_join", "mcs_ctx_set_async_search", "mcs_ctx_search_fence", "mcs_search_kf_kf", "mcs_search_kf_f", "mcs_search_triangulation", "mcs_search_kf_f_sweep",
    "mcs_search_triangulation_sweep", "mcs_search_kf_kf_ring", "mcs_rows_valid", "mcs_extractor_set_describe", "mcs_extractor_describe_stats", "mcs_extractor_tie_stats", "mcs_extractor_set_tie_band", "mcs_extractor_fix_ties", "mcs_extractor_tie_counts", "mcs_extractor_set_tie_capture", "mcs_extractor_patch_ties", "mcs_describe_fast_bound",
    "mcs_selftest_describe_fast", "mcs_describe_fast_table", "mcs_describe_fast_table_packed", "mcs_extract_batch_strided", "mcs_rig_pack_headers", "mcs_rig_rows_valid",
    "mcs_search_by_projection", "mcs_window_match", "mcs_window_best", "mcs_rotation_consistency", "mcs_world_to_cam", "mcs_distinctive_descriptors", "mcs_selftest_shared_reciprocal", "mcs_vocabulary_create", "mcs_vocabulary_destroy", "mcs_bow_transform", "mcs_copy_narrow", "mcs_ctx_result_stream", "mcs_ctx_stream_conflicts", "mcs_ctx_transfer_stream", "mcs_host_alloc", "mcs_host_free",
]

WINDOW_RATIO, WINDOW_BEST, WINDOW_INITIALIZE = 1, 2, 3


class WindowProbes(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("radius", C.c_void_p), ("min_level", C.c_void_p), ("max_level", C.c_void_p),
                ("cam", C.c_void_p), ("desc", C.c_void_p), ("mask", C.c_void_p), ("n", C.c_int32), ("stride", C.c_int32), ("accepted_out", C.c_void_p)]


class ProjectionSet(C.Structure):
    _fields_ = [("proj_x", C.c_void_p), ("proj_y", C.c_void_p), ("view_cos", C.c_void_p), ("level", C.c_void_p), ("cam", C.c_void_p),
                ("desc", C.c_void_p), ("mask", C.c_void_p), ("n", C.c_int32), ("stride", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("keys", C.c_void_p), ("desc", C.c_void_p), ("mask", C.c_void_p), ("cam", C.c_void_p), ("assigned", C.c_void_p),
                ("n", C.c_int32), ("stride", C.c_int32), ("nr_cams", C.c_int32), ("width", C.c_void_p), ("height", C.c_void_p),
                ("scale_factors", C.c_void_p), ("nlevels", C.c_int32)]

_lib = None


def lib():
    """Load libmcs_hip.so (built in-tree by __graft_entry__.build() / make -C csrc).  Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `make -C multicol-slam_amd/csrc` (hipcc, gfx950). "
                          "There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
    L.mcs_last_error.restype = C.c_char_p
    L.mcs_device_count.argtypes = [i32p]
    L.mcs_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.mcs_ctx_destroy.argtypes = [vp]
    L.mcs_ctx_synchronize.argtypes = [vp]
    L.mcs_ctx_join.argtypes = [vp]
    L.mcs_ctx_set_async_search.argtypes = [vp, C.c_int]
    L.mcs_ctx_search_fence.argtypes = [vp, C.c_int]
    L.mcs_ctx_enable_timing.argtypes = [vp, C.c_int]
    L.mcs_ctx_kernel_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float)]
    L.mcs_extractor_create.argtypes = [vp, C.POINTER(ExtractorParams), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.mcs_extractor_destroy.argtypes = [vp]
    L.mcs_extractor_kp_capacity.argtypes = [vp, i32p]
    L.mcs_extractor_levels.argtypes = [vp, i32p, i32p, i32p, i32p]
    L.mcs_extractor_status.argtypes = [vp]
    L.mcs_extractor_set_describe.argtypes = [vp, C.c_int, C.c_double]
    L.mcs_extractor_describe_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.mcs_extractor_tie_stats.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
    L.mcs_extractor_set_tie_band.argtypes = [vp, C.c_double]
    L.mcs_extractor_set_masks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    L.mcs_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.mcs_host_free.argtypes = [vp, vp]
    L.mcs_extractor_fix_ties.argtypes = [vp, C.POINTER(C.c_int)]
    L.mcs_extractor_set_tie_capture.argtypes = [vp, C.c_int, C.c_int]
    L.mcs_extractor_patch_ties.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mcs_extractor_tie_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.mcs_describe_fast_bound.argtypes = [C.POINTER(Ocam), C.c_int, C.POINTER(C.c_double)]
    L.mcs_selftest_describe_fast.argtypes = [vp, C.POINTER(Ocam), C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    L.mcs_describe_fast_table.argtypes = [C.POINTER(Ocam), vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), vp]
    L.mcs_describe_fast_table_packed.argtypes = [C.POINTER(Ocam), vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.mcs_extract_batch.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]
    L.mcs_extract_batch_strided.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_int, vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_int]
    L.mcs_rig_pack_headers.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int]
    L.mcs_rig_rows_valid.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.mcs_extractor_tap_level.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    L.mcs_extractor_tap_candidates.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, i32p]
    L.mcs_extractor_tap_selected.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, i32p]
    L.mcs_match_topk.argtypes = [vp, C.POINTER(DescSet), C.POINTER(DescSet), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.mcs_match_topk_batched.argtypes = [vp, C.c_int, C.POINTER(DescSet), C.c_size_t, C.POINTER(DescSet), C.c_size_t, C.c_int,
                                         C.c_int, C.c_int, C.c_int, vp, vp, vp]
    srch = [vp, C.c_int, C.POINTER(DescSet), C.c_size_t, C.POINTER(DescSet), C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, vp, vp, vp]
    L.mcs_search_kf_kf.argtypes = srch
    L.mcs_search_kf_f.argtypes = srch
    L.mcs_search_triangulation.argtypes = [vp, C.c_int, C.POINTER(DescSet), C.c_size_t, C.POINTER(DescSet), C.c_size_t, vp, vp, vp, C.c_int,
                                           C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.mcs_search_kf_kf_ring.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(DescSet), C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, vp, vp, vp]
    L.mcs_search_kf_f_sweep.argtypes = [vp, C.c_int, C.POINTER(DescSet), C.c_size_t, C.c_int, C.POINTER(DescSet), C.c_size_t, C.c_int, C.c_double, C.c_int,
                                        C.c_int, vp, vp, vp]
    L.mcs_search_triangulation_sweep.argtypes = [vp, C.c_int, C.POINTER(DescSet), C.c_size_t, C.POINTER(DescSet), C.c_size_t, vp, vp, vp, C.c_size_t,
                                                 C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.mcs_rows_valid.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.mcs_search_by_projection.argtypes = [vp, C.POINTER(ProjectionSet), C.POINTER(FrameView), C.c_double, C.c_double, C.c_int, C.c_int, vp, vp]
    L.mcs_window_match.argtypes = [vp, C.POINTER(WindowProbes), C.POINTER(FrameView), C.c_int, C.c_double, C.c_int, C.c_int, vp, vp]
    L.mcs_window_best.argtypes = [vp, C.POINTER(WindowProbes), C.POINTER(FrameView), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.mcs_rotation_consistency.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.mcs_world_to_cam.argtypes = [vp, vp, C.POINTER(Ocam), C.c_int, C.POINTER(vp), vp, vp, C.c_int, C.c_int, vp, vp]
    L.mcs_distinctive_descriptors.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]
    L.mcs_selftest_shared_reciprocal.argtypes = [vp, C.c_uint64, C.c_int, i32p]
    L.mcs_vocabulary_create.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(vp)]
    L.mcs_vocabulary_destroy.argtypes = [vp]
    L.mcs_vocabulary_destroy.restype = None
    L.mcs_bow_transform.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.mcs_copy_narrow.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp]
    L.mcs_ctx_result_stream.argtypes = [vp, C.POINTER(vp)]
    L.mcs_ctx_stream_conflicts.argtypes = [vp, vp, C.POINTER(C.c_uint)]
    L.mcs_ctx_transfer_stream.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint)]
    L.mcs_descriptor_distance.argtypes = [vp, vp, vp, C.c_int, i32p]
    L.mcs_descriptor_distance_masked.argtypes = [vp, vp, vp, vp, vp, C.c_int, i32p]
    _lib = L
    return L


def check(rc):
    if rc != MCS_OK:
        raise McsError(rc, lib().mcs_last_error().decode("utf-8", "replace"))


def np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
