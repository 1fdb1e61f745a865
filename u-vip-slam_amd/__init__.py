"""uvo -- MI355X-native ORB feature front-end (extract + match), Python plumbing over the C ABI.

The product is `libuvo.so` (hand-written HIP for gfx950 behind include/uvo/uvo.h).  This module only loads it with
ctypes and mirrors the reference's two entry classes (USLAM::ORBextractor, include/ORBextractor.h:47-95;
USLAM::ORBmatcher, include/ORBmatcher.h:41-88) so the parity tests and bench.py read like calls into the
reference.  There is no CPU fallback: importing fails loudly when the library is missing, and creating an
extractor / matcher fails when no HIP device is usable.

The package directory is not a valid Python identifier; import it with
    importlib.import_module("u-vip-slam_amd")
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuvo.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
                           ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

UVO_OK, UVO_E_BADARG, UVO_E_NODEVICE, UVO_E_HIP, UVO_E_CAPACITY, UVO_E_UNSUPPORTED, UVO_E_NOMEM = 0, -1, -2, -3, -4, -5, -6
UVO_TUNE_OCT_WIDE_MAX = 1
UVO_TUNE_FAST_MODE = 2
UVO_FAST_MODE_ADAPTIVE, UVO_FAST_MODE_TWO_PASS, UVO_FAST_MODE_SINGLE_PASS = 0, 1, 2
UVO_TUNE_BLUR_ROUNDING, UVO_BLUR_ROUNDING_SCALAR, UVO_BLUR_ROUNDING_SSE2 = 9, 0, 1
UVO_TUNE_LEVEL0_INPLACE = 11
# launch-shape knobs outside the public header (csrc/tune_internal.h): the parity tests force each shape through them
UVO_TUNE_FUSE_BLUR_TREE = 10
UVO_TUNE_PYR_RING = 12
UVO_TUNE_PYR_FORM, UVO_PYR_FORM_AUTO, UVO_PYR_FORM_LEVELS, UVO_PYR_FORM_TILES = 13, 0, 1, 2
UVO_TUNE_PYR_TILE_GROUP = 14
UVO_TUNE_FEW_FRAMES = 16
UVO_TUNE_ZERO_COPY_OUT, UVO_TUNE_SPIN_WAIT = 17, 18

# every symbol include/uvo/uvo.h declares
ABI_SYMBOLS = [
    "uvo_extractor_create", "uvo_extractor_destroy", "uvo_extractor_levels", "uvo_extractor_max_keypoints", "uvo_extractor_scale_factor", "uvo_extractor_tables",
    "uvo_extract", "uvo_extract_tracked", "uvo_extract_batch", "uvo_extract_batch_device", "uvo_host_alloc", "uvo_host_free", "uvo_host_register", "uvo_host_unregister", "uvo_host_bind_near_device", "uvo_host_bind_to_cpulist_file", "uvo_shard_plan_make", "uvo_sharder_create", "uvo_sharder_destroy", "uvo_sharder_max_keypoints", "uvo_sharder_run", "uvo_sharder_submit", "uvo_sharder_wait", "uvo_extract_batch_submit", "uvo_extract_batch_wait", "uvo_extractor_synchronize", "uvo_extractor_set_pipeline", "uvo_extractor_tune", "uvo_extractor_fast_state", "uvo_extractor_level_dims",
    "uvo_grider_fast", "uvo_clahe", "uvo_clahe_batch_device", "uvo_extractor_read_plane", "uvo_extractor_read_candidates", "uvo_extractor_profile", "uvo_extractor_profile_only", "uvo_extractor_kernel_times",
    "uvo_matcher_create", "uvo_matcher_destroy", "uvo_matcher_synchronize", "uvo_hamming_knn2", "uvo_hamming_knn2_batch_device",
    "uvo_hamming_matrix", "uvo_distinctive_descriptors", "uvo_search_by_projection", "uvo_match_windows", "uvo_match_groups",
    "uvo_search_by_projection_kf", "uvo_search_by_bow", "uvo_search_for_triangulation", "uvo_search_for_triangulation_batch", "uvo_search_for_triangulation_next", "uvo_fuse", "uvo_fuse_batch", "uvo_project_points", "uvo_search_points_in_frustum", "uvo_sim3_decompose", "uvo_sim3_relative", "uvo_project_sim3", "uvo_search_by_projection_sim3", "uvo_search_by_sim3", "uvo_haloc_hash", "uvo_klt_create", "uvo_klt_destroy", "uvo_klt_build_pyramid", "uvo_klt_build_pyramid_from_extractor", "uvo_klt_read_level", "uvo_klt_track", "uvo_undistort_points", "uvo_klt_track_undistorted", "uvo_vocabulary_create", "uvo_vocabulary_destroy", "uvo_bow_transform", "uvo_matcher_wait_extractor", "uvo_extractor_wait_matcher", "uvo_matcher_attach_extractor", "uvo_matcher_profile",
    "uvo_matcher_kernel_times", "uvo_last_error", "uvo_device_info",
]


class MatchRule(ctypes.Structure):
    """uvo_match_rule (include/uvo/uvo.h)."""
    _fields_ = [("rule", ctypes.c_int32), ("max_dist", ctypes.c_int32), ("nn_ratio", ctypes.c_float), ("exclusive", ctypes.c_int32),
                ("check_orientation", ctypes.c_int32)]


class FeatureVectorC(ctypes.Structure):
    """uvo_feature_vector: a DBoW2::FeatureVector in flat form."""
    _fields_ = [("node", ctypes.c_void_p), ("start", ctypes.c_void_p), ("feat", ctypes.c_void_p), ("n_nodes", ctypes.c_int32)]


class Epipolar(ctypes.Structure):
    """uvo_epipolar."""
    _fields_ = [("f12", ctypes.c_float * 9), ("q_x", ctypes.c_void_p), ("q_y", ctypes.c_void_p), ("t_x", ctypes.c_void_p), ("t_y", ctypes.c_void_p),
                ("sigma2", ctypes.c_void_p), ("nlevels", ctypes.c_int32)]


class CameraPose(ctypes.Structure):
    """uvo_camera_pose."""
    _fields_ = [("rcw", ctypes.c_float * 9), ("tcw", ctypes.c_float * 3), ("ow", ctypes.c_float * 3), ("fx", ctypes.c_float), ("fy", ctypes.c_float),
                ("cx", ctypes.c_float), ("cy", ctypes.c_float), ("min_x", ctypes.c_float), ("max_x", ctypes.c_float), ("min_y", ctypes.c_float),
                ("max_y", ctypes.c_float)]

    @classmethod
    def make(cls, Rcw, tcw, Ow, fx, fy, cx, cy, bounds):
        """bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY)"""
        c = cls()
        c.rcw[:] = [float(x) for x in np.asarray(Rcw, np.float32).reshape(9)]
        c.tcw[:] = [float(x) for x in np.asarray(tcw, np.float32).reshape(3)]
        c.ow[:] = [float(x) for x in np.asarray(Ow, np.float32).reshape(3)]
        c.fx, c.fy, c.cx, c.cy = float(fx), float(fy), float(cx), float(cy)
        c.min_x, c.min_y, c.max_x, c.max_y = [float(b) for b in bounds]
        return c

    def as_array(self):
        return np.frombuffer(bytes(self), np.float32).copy()


class TriangulationPairC(ctypes.Structure):
    """uvo_triangulation_pair."""
    _fields_ = [("fv2", ctypes.c_void_p), ("kp2", ctypes.c_void_p), ("n2", ctypes.c_int32), ("desc2", ctypes.c_void_p), ("has_mp2", ctypes.c_void_p),
                ("f12", ctypes.c_float * 9), ("sigma2", ctypes.c_void_p), ("nlevels", ctypes.c_int32)]


class FuseTargetC(ctypes.Structure):
    """uvo_fuse_target."""
    _fields_ = [("kp", ctypes.c_void_p), ("n", ctypes.c_int32), ("desc", ctypes.c_void_p), ("min_x", ctypes.c_int32), ("min_y", ctypes.c_int32),
                ("max_x", ctypes.c_int32), ("max_y", ctypes.c_int32), ("cam", CameraPose), ("scale_factors", ctypes.c_void_p), ("nlevels", ctypes.c_int32)]


PROJECT_FRUSTUM, PROJECT_KF_RELOC, PROJECT_FUSE, PROJECT_PIXEL_BOUNDED, PROJECT_PIXEL = range(5)
RULE_BEST_RATIO_SAME_LEVEL, RULE_BEST_ONLY, RULE_BEST_RATIO_LE, RULE_BEST_RATIO_LT, RULE_TRIANGULATION, RULE_BEST_RATIO_LEQ, RULE_INIT_STEAL = range(7)


class FeatureVector:
    """DBoW2::FeatureVector stand-in: {node id: [feature indices]} kept as the flat arrays the C ABI takes."""

    def __init__(self, groups):
        nodes = sorted(groups)
        self.node = np.asarray(nodes, np.uint32)
        self.start = np.zeros(len(nodes) + 1, np.int32)
        feats = []
        for j, k in enumerate(nodes):
            feats.extend(int(v) for v in groups[k])
            self.start[j + 1] = len(feats)
        self.feat = np.asarray(feats, np.int32)
        self.c = FeatureVectorC(_ptr(self.node), _ptr(self.start), _ptr(self.feat) if len(feats) else None, len(nodes))


class UvoError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed with code %d: %s" % (where, code, last_error()))


class ExtractorCfg(ctypes.Structure):
    _fields_ = [("nfeatures", ctypes.c_int32), ("scale_factor", ctypes.c_float), ("nlevels", ctypes.c_int32), ("score_type", ctypes.c_int32),
                ("fast_th", ctypes.c_int32), ("max_width", ctypes.c_int32), ("max_height", ctypes.c_int32), ("max_batch", ctypes.c_int32),
                ("max_input_keypoints", ctypes.c_int32), ("device", ctypes.c_int32)]


UVO_SHARD_MAX, UVO_SHARD_REMOTE = 64, -1


class SharderCfg(ctypes.Structure):
    _fields_ = [("extractor", ExtractorCfg), ("n_shards", ctypes.c_int32), ("devices", ctypes.c_int32 * UVO_SHARD_MAX), ("chunk_frames", ctypes.c_int32),
                ("match", ctypes.c_int32)]


class ShardPlan(ctypes.Structure):
    _fields_ = [("first_frame", ctypes.c_int32), ("n_frames", ctypes.c_int32), ("first_pair", ctypes.c_int32), ("n_pairs", ctypes.c_int32),
                ("halo_frame", ctypes.c_int32), ("n_chunks", ctypes.c_int32)]


class MatcherCfg(ctypes.Structure):
    _fields_ = [("max_query", ctypes.c_int32), ("max_train", ctypes.c_int32), ("max_batch", ctypes.c_int32), ("max_map_points", ctypes.c_int32),
                ("device", ctypes.c_int32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libuvo.so not built (%s); run `python u-vip-slam_amd/build.py` -- there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cf, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_ssize_t
    lib.uvo_last_error.restype = ctypes.c_char_p
    lib.uvo_device_info.argtypes = [ci, ctypes.c_char_p, ci]
    lib.uvo_extractor_create.argtypes = [ctypes.POINTER(ExtractorCfg), ctypes.POINTER(vp)]
    lib.uvo_extractor_destroy.argtypes = [vp]
    lib.uvo_extractor_destroy.restype = None
    lib.uvo_extractor_levels.argtypes = [vp]
    lib.uvo_extractor_max_keypoints.argtypes = [vp]
    lib.uvo_extractor_scale_factor.argtypes = [vp]
    lib.uvo_extractor_scale_factor.restype = cf
    lib.uvo_extractor_tables.argtypes = [vp, vp, vp, vp, vp]
    lib.uvo_extract.argtypes = [vp, vp, ci, ci, cl, vp, ci, vp, ci, ci, ci, ci, ci, vp, vp, ci, vp]
    lib.uvo_extract_tracked.argtypes = [vp, vp, ci, ci, cl, vp, ci, ci, ci, vp, vp, ci, vp, vp]
    lib.uvo_extract_batch.argtypes = [vp, ci, vp, ci, ci, cl, cl, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, ci, vp]
    lib.uvo_extract_batch_device.argtypes = [vp, ci, vp, ci, ci, cl, cl, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, ci, vp]
    lib.uvo_host_alloc.argtypes = [vp, ctypes.c_size_t]
    lib.uvo_host_free.argtypes = [vp]
    lib.uvo_host_register.argtypes = [vp, ctypes.c_size_t]
    lib.uvo_host_unregister.argtypes = [vp]
    lib.uvo_host_bind_near_device.argtypes = [ci, vp]
    lib.uvo_host_bind_to_cpulist_file.argtypes = [ctypes.c_char_p]
    lib.uvo_shard_plan_make.argtypes = [ci, ci, ci, ci, ctypes.POINTER(ShardPlan)]
    lib.uvo_sharder_create.argtypes = [ctypes.POINTER(SharderCfg), ctypes.POINTER(vp)]
    lib.uvo_sharder_destroy.argtypes = [vp]
    lib.uvo_sharder_destroy.restype = None
    lib.uvo_sharder_max_keypoints.argtypes = [vp]
    lib.uvo_sharder_run.argtypes = [vp, vp, ci, ci, ci, ci, ci, cl, cl, vp, vp, ci, vp, vp, vp, vp, vp]
    lib.uvo_sharder_submit.argtypes = [vp, vp, ci, ci, ci, ci, ci, cl, cl, vp, vp, ci, vp, vp, vp, vp, vp, vp]
    lib.uvo_sharder_wait.argtypes = [vp, ci]
    lib.uvo_extract_batch_submit.argtypes = [vp, ci, vp, ci, ci, cl, cl, vp, vp, ci, vp, vp]
    lib.uvo_extract_batch_wait.argtypes = [vp, ci]
    lib.uvo_extractor_synchronize.argtypes = [vp]
    lib.uvo_extractor_set_pipeline.argtypes = [vp, ci]
    lib.uvo_extractor_tune.argtypes = [vp, ci, ci]
    lib.uvo_extractor_level_dims.argtypes = [vp, ci, vp, vp]
    lib.uvo_extractor_fast_state.argtypes = [vp, vp, vp, vp]
    lib.uvo_clahe.argtypes = [vp, vp, ci, ci, cl, ctypes.c_double, ci, ci, vp, cl]
    lib.uvo_clahe_batch_device.argtypes = [vp, ci, vp, ci, ci, cl, cl, ctypes.c_double, ci, ci, vp, cl, cl]
    lib.uvo_grider_fast.argtypes = [vp, vp, ci, ci, cl, ci, ci, ci, ci, ci, vp, ci, vp]
    lib.uvo_extractor_read_plane.argtypes = [vp, ci, ci, ci, vp]
    lib.uvo_extractor_read_candidates.argtypes = [vp, ci, ci, vp, ci, vp]
    lib.uvo_extractor_profile.argtypes = [vp, ci]
    lib.uvo_extractor_profile_only.argtypes = [vp, ctypes.c_char_p]
    lib.uvo_extractor_kernel_times.argtypes = [vp, ctypes.c_char_p, ci, vp, vp, ci, vp]
    lib.uvo_matcher_create.argtypes = [ctypes.POINTER(MatcherCfg), ctypes.POINTER(vp)]
    lib.uvo_matcher_destroy.argtypes = [vp]
    lib.uvo_matcher_destroy.restype = None
    lib.uvo_matcher_synchronize.argtypes = [vp]
    lib.uvo_hamming_knn2.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp, vp, vp]
    lib.uvo_hamming_knn2_batch_device.argtypes = [vp, ci, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp]
    lib.uvo_hamming_matrix.argtypes = [vp, vp, ci, vp, ci, vp]
    lib.uvo_distinctive_descriptors.argtypes = [vp, vp, vp, ci, vp, vp]
    lib.uvo_search_by_projection.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp]
    lib.uvo_match_windows.argtypes = [vp, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.uvo_match_groups.argtypes = [vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.uvo_search_by_projection_kf.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, ci, ci, vp]
    lib.uvo_search_by_bow.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, cf, ci, vp, vp]
    lib.uvo_search_for_triangulation.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, vp, vp]
    lib.uvo_project_points.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp, vp, vp, vp, vp]
    lib.uvo_vocabulary_create.argtypes = [vp, ctypes.POINTER(vp)]
    lib.uvo_vocabulary_destroy.argtypes = [vp]
    lib.uvo_vocabulary_destroy.restype = None
    lib.uvo_bow_transform.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, vp]
    lib.uvo_haloc_hash.argtypes = [vp, vp, ci, ci, vp, ci, vp]
    lib.uvo_klt_create.argtypes = [vp, ctypes.POINTER(vp)]
    lib.uvo_klt_destroy.argtypes = [vp]
    lib.uvo_klt_destroy.restype = None
    lib.uvo_klt_build_pyramid.argtypes = [vp, ci, vp, ci, ci, cl, vp]
    lib.uvo_klt_build_pyramid_from_extractor.argtypes = [vp, ci, vp, vp]
    lib.uvo_klt_read_level.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    lib.uvo_klt_track.argtypes = [vp, ci, ci, vp, vp, ci, ci, ci, ctypes.c_double, ctypes.c_double, vp, vp]
    lib.uvo_undistort_points.argtypes = [vp, vp, vp, ci, vp]
    lib.uvo_klt_track_undistorted.argtypes = [vp, ci, ci, vp, vp, ci, ci, ci, ctypes.c_double, ctypes.c_double, vp, vp, vp, vp, vp]
    lib.uvo_fuse.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, cf, vp, vp]
    lib.uvo_search_for_triangulation_batch.argtypes = [vp, vp, vp, ci, vp, vp, ci, vp]
    lib.uvo_search_for_triangulation_next.argtypes = [vp, ci, vp, ci, vp, vp]
    lib.uvo_fuse_batch.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, cf, vp, vp]
    lib.uvo_search_points_in_frustum.argtypes = [vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf, cf, cf, vp, vp, vp, vp, vp, vp, vp]
    lib.uvo_sim3_decompose.argtypes = [vp, ci, vp]
    lib.uvo_sim3_relative.argtypes = [cf, vp, vp, vp, vp, vp]
    lib.uvo_project_sim3.argtypes = [vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp]
    lib.uvo_search_by_projection_sim3.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, vp]
    lib.uvo_search_by_sim3.argtypes = [vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, ci, cf, vp, vp]
    lib.uvo_matcher_wait_extractor.argtypes = [vp, vp]
    lib.uvo_extractor_wait_matcher.argtypes = [vp, vp]
    lib.uvo_matcher_attach_extractor.argtypes = [vp, vp]
    lib.uvo_matcher_profile.argtypes = [vp, ci]
    lib.uvo_matcher_kernel_times.argtypes = [vp, ctypes.c_char_p, ci, vp, vp, ci, vp]
    return lib


lib = _load()


def last_error():
    return lib.uvo_last_error().decode("utf-8", "replace")


def device_info(device=0):
    buf = ctypes.create_string_buffer(256)
    rc = lib.uvo_device_info(device, buf, 256)
    if rc:
        raise UvoError(rc, "uvo_device_info")
    return buf.value.decode()


def _ptr(a):
    return None if a is None else a.ctypes.data


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (uvo_host_alloc), for the asynchronous host-buffer form; freed with the array."""
    import weakref
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = ctypes.c_void_p()
    rc = lib.uvo_host_alloc(ctypes.byref(p), max(n, 1))
    if rc:
        raise UvoError(rc, "uvo_host_alloc")
    buf = (ctypes.c_uint8 * max(n, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(buf, lib.uvo_host_free, p.value)
    return arr


def shard_plan(total_frames, n_shards, shard, chunk_frames):
    """uvo_shard_plan_make: the block of one shard (pure arithmetic, works without a GPU)."""
    p = ShardPlan()
    rc = lib.uvo_shard_plan_make(total_frames, n_shards, shard, chunk_frames, ctypes.byref(p))
    if rc:
        raise UvoError(rc, "uvo_shard_plan_make")
    return p


class Sharder:
    """uvo_sharder: one job of `total` frames over the GPUs of one node, results gathered by the device-to-host copies themselves
    (include/uvo/uvo.h "Sharder").  devices[i] = HIP ordinal of shard i, or UVO_SHARD_REMOTE for shards another process runs."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, fastTh=20, *, max_width=640, max_height=512, devices=(0,), chunk_frames=128,
                 match=True):
        self.cfg = SharderCfg()
        self.cfg.extractor = ExtractorCfg(nfeatures, scaleFactor, nlevels, 0, fastTh, max_width, max_height, 1, 0, 0)
        self.cfg.n_shards = len(devices)
        for i, d in enumerate(devices):
            self.cfg.devices[i] = d
        self.cfg.chunk_frames = chunk_frames
        self.cfg.match = 1 if match else 0
        self._h = ctypes.c_void_p()
        rc = lib.uvo_sharder_create(ctypes.byref(self.cfg), ctypes.byref(self._h))
        if rc:
            self._h = None
            raise UvoError(rc, "uvo_sharder_create")
        self.cap = lib.uvo_sharder_max_keypoints(self._h)
        self.n_shards, self.chunk_frames = len(devices), chunk_frames

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.uvo_sharder_destroy(self._h)
            self._h = None

    __del__ = close

    def plan(self, total_frames, shard):
        return shard_plan(total_frames, self.n_shards, shard, self.chunk_frames)

    def _check(self, imgs, total_frames, out_kp, out_desc, n_out, idx0, d0, idx1, d1):
        assert imgs.dtype == np.uint8 and imgs.ndim == 3 and imgs.flags.c_contiguous
        cap = out_kp.shape[1]
        assert out_kp.dtype == KEYPOINT_DTYPE and out_kp.shape[0] >= total_frames and out_kp.flags.c_contiguous
        assert out_desc.dtype == np.uint8 and out_desc.shape[1:] == (cap, 32) and out_desc.flags.c_contiguous
        assert n_out.dtype == np.int32 and len(n_out) >= total_frames
        if idx0 is not None:
            for a, dt in ((idx0, np.int32), (idx1, np.int32), (d0, np.uint16), (d1, np.uint16)):
                assert a.dtype == dt and a.shape[1] == cap and a.shape[0] >= total_frames - 1 and a.flags.c_contiguous
        return cap

    def submit(self, imgs, imgs_first_frame, total_frames, out_kp, out_desc, n_out, idx0=None, d0=None, idx1=None, d1=None):
        """uvo_sharder_submit: queue a job, return its ticket at once.  imgs: (n, H, W) uint8 C-contiguous holding frames
        imgs_first_frame .. ; outputs as uvo_sharder_submit documents them (untouched until wait(ticket) returns)."""
        cap = self._check(imgs, total_frames, out_kp, out_desc, n_out, idx0, d0, idx1, d1)
        _, h, w = imgs.shape
        t = ctypes.c_int()
        rc = lib.uvo_sharder_submit(self._h, imgs.ctypes.data, imgs.shape[0], imgs_first_frame, total_frames, w, h, w, w * h, out_kp.ctypes.data, out_desc.ctypes.data, cap,
                                    n_out.ctypes.data, _ptr(idx0), _ptr(d0), _ptr(idx1), _ptr(d1), ctypes.byref(t))
        if rc:
            raise UvoError(rc, "uvo_sharder_submit")
        return t.value

    def wait(self, ticket):
        rc = lib.uvo_sharder_wait(self._h, int(ticket))
        if rc:
            raise UvoError(rc, "uvo_sharder_wait")

    def run(self, imgs, imgs_first_frame, total_frames, out_kp, out_desc, n_out, idx0=None, d0=None, idx1=None, d1=None):
        """submit + wait."""
        self.wait(self.submit(imgs, imgs_first_frame, total_frames, out_kp, out_desc, n_out, idx0, d0, idx1, d1))


def host_register(arr):
    """Page-lock memory the caller owns (uvo_host_register), e.g. a mapping shared between processes."""
    rc = lib.uvo_host_register(arr.ctypes.data, arr.nbytes)
    if rc:
        raise UvoError(rc, "uvo_host_register")


def host_bind_near_device(device):
    """uvo_host_bind_near_device: binds the calling thread to the CPUs local to `device`; -> (bound, numa_node) -- (False, -1) where the
    platform says nothing (placement is speed, never correctness)."""
    node = ctypes.c_int32(-1)
    rc = lib.uvo_host_bind_near_device(int(device), ctypes.byref(node))
    if rc < 0:
        raise UvoError(rc, "uvo_host_bind_near_device")
    return rc == 1, int(node.value)


def host_bind_to_cpulist_file(path):
    rc = lib.uvo_host_bind_to_cpulist_file(str(path).encode())
    if rc < 0:
        raise UvoError(rc, "uvo_host_bind_to_cpulist_file")
    return rc == 1


def host_unregister(arr):
    rc = lib.uvo_host_unregister(arr.ctypes.data)
    if rc:
        raise UvoError(rc, "uvo_host_unregister")


def _spread_rows(names, ms):
    """the "kernel:stat" pseudo-rows of a uvo_*_kernel_times report -> {kernel: {stat: ms}}"""
    out = {}
    for i, nm in enumerate(names):
        if ":" in nm:
            k, stat = nm.split(":", 1)
            out.setdefault(k, {})[stat] = float(ms[i])
    return out


class ORBextractor:
    """Mirror of USLAM::ORBextractor (include/ORBextractor.h:47-95).

    ctor args follow the reference (nfeatures, scaleFactor, nlevels, scoreType, fastTh); the extra keyword
    arguments size the device scratch the handle owns.
    """
    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, scoreType=0, fastTh=7, *, max_width=640, max_height=512, max_batch=1,
                 max_input_keypoints=0, device=0):
        self.cfg = ExtractorCfg(nfeatures, scaleFactor, nlevels, scoreType, fastTh, max_width, max_height, max_batch, max_input_keypoints,
                                device)
        self._h = ctypes.c_void_p()
        rc = lib.uvo_extractor_create(ctypes.byref(self.cfg), ctypes.byref(self._h))
        if rc:
            self._h = None
            raise UvoError(rc, "uvo_extractor_create")
        scale = np.zeros(nlevels, np.float32)
        inv = np.zeros(nlevels, np.float32)
        quota = np.zeros(nlevels, np.int32)
        umax = np.zeros(16, np.int32)
        lib.uvo_extractor_tables(self._h, _ptr(scale), _ptr(inv), _ptr(quota), _ptr(umax))
        self.mvScaleFactor, self.mvInvScaleFactor, self.mnFeaturesPerLevel, self.umax = scale, inv, quota, umax
        # upper bound of keypoints per frame: sum(quota + 4) + pass-through keypoints
        self.cap = lib.uvo_extractor_max_keypoints(self._h)
        if self.cap < 0:
            raise UvoError(self.cap, "uvo_extractor_max_keypoints")

    def close(self):
        if getattr(self, "_h", None) and lib is not None:  # `lib` is already gone when the interpreter tears the module down
            lib.uvo_extractor_destroy(self._h)
            self._h = None

    __del__ = close

    def GetLevels(self):
        return lib.uvo_extractor_levels(self._h)

    def GetScaleFactor(self):
        return lib.uvo_extractor_scale_factor(self._h)

    def __call__(self, image, keypoints=None, grid_2d=None, min_px_dist=20, FullDetect=True, num_featsneeded=0):
        """operator()(image, mask, keypoints, descriptors, grid_2d, min_px_dist, FullDetect, num_featsneeded).

        image: HxW uint8.  keypoints: KEYPOINT_DTYPE array (the reference's in/out vector on entry) or None.
        grid_2d: int32 array of shape (rows, cols) in Fortran (column-major) order, mutated in place like the
        reference's Eigen::MatrixXi&.  Returns (keypoints, descriptors).
        """
        if image is None:      # the handle's last clahe() result, already in HBM
            h, w = self._clahe_shape
        else:
            image = np.ascontiguousarray(image, dtype=np.uint8)
            h, w = image.shape
        n_in = 0 if keypoints is None else len(keypoints)
        kin = None if n_in == 0 else np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE)
        rows = cols = 0
        if grid_2d is not None:
            if not (grid_2d.dtype == np.int32 and grid_2d.flags.f_contiguous):
                raise ValueError("grid_2d must be int32 and column-major (np.asfortranarray)")
            rows, cols = grid_2d.shape
        out_kp = np.zeros(self.cap, KEYPOINT_DTYPE)
        out_desc = np.zeros((self.cap, 32), np.uint8)
        n_out = ctypes.c_int(0)
        rc = lib.uvo_extract(self._h, _ptr(image), w, h, w if image is None else image.strides[0], _ptr(kin), n_in, _ptr(grid_2d), rows, cols, int(min_px_dist),
                             1 if FullDetect else 0, int(num_featsneeded), out_kp.ctypes.data, out_desc.ctypes.data, self.cap,
                             ctypes.byref(n_out))
        if rc:
            raise UvoError(rc, "uvo_extract")
        n = n_out.value
        return out_kp[:n].copy(), out_desc[:n].copy()

    def extract_tracked(self, image, keypoints, min_px_dist, num_featsneeded, want_grid=False):
        """uvo_extract_tracked: src/Tracking.cc:896-946 as one call -- occupancy grid from the tracked keypoints on the device, then the
        top-up extraction.  image None = the last clahe() result.  Returns (keypoints, descriptors[, grid (rows, cols) int32 F-order])."""
        if image is None:
            h, w = self._clahe_shape
        else:
            image = np.ascontiguousarray(image, dtype=np.uint8)
            h, w = image.shape
        kin = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE)
        out_kp, out_desc, n_out = np.zeros(self.cap, KEYPOINT_DTYPE), np.zeros((self.cap, 32), np.uint8), ctypes.c_int(0)
        grid = np.zeros((h // min_px_dist + 2, w // min_px_dist + 2), np.int32, order="F") if want_grid else None
        rc = lib.uvo_extract_tracked(self._h, _ptr(image), w, h, w if image is None else image.strides[0], _ptr(kin) if len(kin) else None, len(kin),
                                     int(min_px_dist), int(num_featsneeded), out_kp.ctypes.data, out_desc.ctypes.data, self.cap, ctypes.byref(n_out),
                                     _ptr(grid))
        if rc:
            raise UvoError(rc, "uvo_extract_tracked")
        n = n_out.value
        return (out_kp[:n].copy(), out_desc[:n].copy()) + ((grid,) if want_grid else ())

    def extract_batch(self, images):
        """FullDetect extraction of a (B, H, W) uint8 stack; returns list of (keypoints, descriptors)."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        b, h, w = images.shape
        out_kp = np.zeros((b, self.cap), KEYPOINT_DTYPE)
        out_desc = np.zeros((b, self.cap, 32), np.uint8)
        n_out = np.zeros(b, np.int32)
        rc = lib.uvo_extract_batch(self._h, b, images.ctypes.data, w, h, images.strides[1], images.strides[0], None, None, None, 0, 0, 0, 1,
                                   None, out_kp.ctypes.data, out_desc.ctypes.data, self.cap, n_out.ctypes.data)
        if rc:
            raise UvoError(rc, "uvo_extract_batch")
        return [(out_kp[i, :n_out[i]].copy(), out_desc[i, :n_out[i]].copy()) for i in range(b)]

    def submit(self, images, out_kp, out_desc, n_out):
        """Asynchronous FullDetect extraction of a (B, H, W) uint8 stack into caller arrays out_kp (B, cap) KEYPOINT_DTYPE,
        out_desc (B, cap, 32) uint8, n_out (B,) int32 -- ideally all from pinned_empty(); returns the ticket for wait()."""
        assert images.dtype == np.uint8 and images.ndim == 3 and images.strides[2] == 1   # rows contiguous; row / frame strides are free
        b, h, w = images.shape
        cap = out_kp.shape[1]
        assert out_kp.dtype == KEYPOINT_DTYPE and out_kp.flags.c_contiguous and out_kp.shape[0] >= b
        assert out_desc.dtype == np.uint8 and out_desc.flags.c_contiguous and out_desc.shape[1:] == (cap, 32)
        assert n_out.dtype == np.int32 and len(n_out) >= b
        t = ctypes.c_int()
        rc = lib.uvo_extract_batch_submit(self._h, b, images.ctypes.data, w, h, images.strides[1], images.strides[0], out_kp.ctypes.data,
                                          out_desc.ctypes.data, cap, n_out.ctypes.data, ctypes.byref(t))
        if rc:
            raise UvoError(rc, "uvo_extract_batch_submit")
        return t.value

    def wait(self, ticket):
        rc = lib.uvo_extract_batch_wait(self._h, int(ticket))
        if rc:
            raise UvoError(rc, "uvo_extract_batch_wait")

    def extract_batch_device(self, d_imgs, batch, width, height, d_out_kp, d_out_desc, d_n_out, cap=None, stride=None, frame_stride=None):
        """HBM-resident FullDetect extraction; all arguments are integer device addresses (rows `stride` bytes apart, frames `frame_stride`;
        tight by default).  Asynchronous: the images must stay as they are until the batch is complete."""
        stride = stride or width
        rc = lib.uvo_extract_batch_device(self._h, batch, d_imgs, width, height, stride, frame_stride or stride * height, None, None, None, 0, 0, 0, 1, None,
                                          d_out_kp, d_out_desc, cap or self.cap, d_n_out)
        if rc:
            raise UvoError(rc, "uvo_extract_batch_device")

    def grider_fast(self, image, num_features, grid_x, grid_y, threshold, nms=True):
        """Grider_FAST::perform_griding (include/Grider_FAST.h:81-137)."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape
        cap = num_features + grid_x * grid_y + 64
        out = np.zeros(cap, KEYPOINT_DTYPE)
        n = ctypes.c_int()
        rc = lib.uvo_grider_fast(self._h, image.ctypes.data, w, h, image.strides[0], num_features, grid_x, grid_y, threshold,
                                 1 if nms else 0, out.ctypes.data, cap, ctypes.byref(n))
        if rc:
            raise UvoError(rc, "uvo_grider_fast")
        return out[:n.value].copy()

    def synchronize(self):
        rc = lib.uvo_extractor_synchronize(self._h)
        if rc:
            raise UvoError(rc, "uvo_extractor_synchronize")

    def set_pipeline(self, depth):
        rc = lib.uvo_extractor_set_pipeline(self._h, depth)
        if rc:
            raise UvoError(rc, "uvo_extractor_set_pipeline")

    def tune(self, knob, value):
        rc = lib.uvo_extractor_tune(self._h, knob, value)
        if rc:
            raise UvoError(rc, "uvo_extractor_tune")

    def fast_state(self):
        """uvo_extractor_fast_state: per level (threshold the next batch of the current lane streams at, fall-back cells counted in the
        last batch, cells per frame)."""
        n = lib.uvo_extractor_levels(self._h)
        t, f, c = (np.zeros(n, np.int32) for _ in range(3))
        rc = lib.uvo_extractor_fast_state(self._h, t.ctypes.data, f.ctypes.data, c.ctypes.data)
        if rc:
            raise UvoError(rc, "uvo_extractor_fast_state")
        return t, f, c

    # ---- stage taps used by the parity tests ----
    def level_dims(self, level):
        w, h = ctypes.c_int(), ctypes.c_int()
        rc = lib.uvo_extractor_level_dims(self._h, level, ctypes.byref(w), ctypes.byref(h))
        if rc:
            raise UvoError(rc, "uvo_extractor_level_dims")
        return w.value, h.value

    def clahe(self, img, clip_limit=4.0, tiles=(12, 12), download=True):
        """cv::CLAHE::apply as set up at src/Tracking.cc:425-431 (clip limit 4, 12 x 12 tiles); returns the enhanced image.
        The result also stays in the handle's HBM: ex(None, ...) extracts from it, KLT.build_pyramid_from(slot, ex) builds the
        optical-flow pyramid from it; download=False skips the copy back (returns None)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        self._clahe_shape = (h, w)
        out = np.empty_like(img) if download else None
        rc = lib.uvo_clahe(self._h, _ptr(img), w, h, img.strides[0], float(clip_limit), int(tiles[0]), int(tiles[1]), _ptr(out),
                           out.strides[0] if download else 0)
        if rc:
            raise UvoError(rc, "uvo_clahe")
        return out

    def clahe_batch_device(self, d_src, batch, w, h, d_dst, clip_limit=4.0, tiles=(12, 12)):
        """HBM-resident form on tight [batch][h][w] buffers (device pointers); enqueued ahead of the next extract_batch_device."""
        rc = lib.uvo_clahe_batch_device(self._h, batch, d_src, w, h, w, w * h, float(clip_limit), int(tiles[0]), int(tiles[1]), d_dst, w, w * h)
        if rc:
            raise UvoError(rc, "uvo_clahe_batch_device")

    def read_plane(self, level, blurred=False, frame=0):
        w, h = self.level_dims(level)
        out = np.zeros((h + 32, w + 32), np.uint8)
        rc = lib.uvo_extractor_read_plane(self._h, frame, level, 1 if blurred else 0, out.ctypes.data)
        if rc:
            raise UvoError(rc, "uvo_extractor_read_plane")
        return out

    def read_candidates(self, level, frame=0, cap=1 << 20):
        buf = np.zeros((cap, 3), np.int32)
        n = ctypes.c_int()
        rc = lib.uvo_extractor_read_candidates(self._h, frame, level, buf.ctypes.data, cap, ctypes.byref(n))
        if rc:
            raise UvoError(rc, "uvo_extractor_read_candidates")
        return buf[:min(n.value, cap)].copy()

    def profile(self, enable=True, only=None):
        """Per-kernel HIP-event timing; only = a kernel name restricts it to that kernel (two event records per launch cost ~3 %)."""
        lib.uvo_extractor_profile_only(self._h, only.encode() if only else None)
        lib.uvo_extractor_profile(self._h, 1 if enable else 0)

    def kernel_times(self):
        """{kernel: (summed ms, launches)}; the per-launch spread rows of the same report ("name:min" ...) are kept in self.last_spread
        as {kernel: {"min" | "p50" | "max" | "period_min" | "period_p50" | "period_max": ms}}."""
        names = ctypes.create_string_buffer(16384)
        ms = np.zeros(256, np.float32)
        launches = np.zeros(256, np.int32)
        n = ctypes.c_int()
        rc = lib.uvo_extractor_kernel_times(self._h, names, 16384, ms.ctypes.data, launches.ctypes.data, 256, ctypes.byref(n))
        if rc:
            raise UvoError(rc, "uvo_extractor_kernel_times")
        nm = names.value.decode().split("\n")[:n.value]
        self.last_spread = _spread_rows(nm, ms)
        return {nm[i]: (float(ms[i]), int(launches[i])) for i in range(n.value) if ":" not in nm[i]}


class ORBmatcher:
    """Mirror of the USLAM::ORBmatcher surface on the hot path (include/ORBmatcher.h:41-88)."""
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio=0.6, checkOri=True, *, max_query=4096, max_train=4096, max_batch=1, max_map_points=8192, device=0):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self.cfg = MatcherCfg(max_query, max_train, max_batch, max_map_points, device)
        self._h = ctypes.c_void_p()
        rc = lib.uvo_matcher_create(ctypes.byref(self.cfg), ctypes.byref(self._h))
        if rc:
            self._h = None
            raise UvoError(rc, "uvo_matcher_create")

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.uvo_matcher_destroy(self._h)
            self._h = None

    __del__ = close

    def knn2(self, q, t, mask=None):
        """All-pairs knn-2 (include/utils.h:100-101).  Returns idx0, d0, idx1, d1."""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq, nt = len(q), len(t)
        idx0 = np.full(nq, -1, np.int32)
        idx1 = np.full(nq, -1, np.int32)
        d0 = np.full(nq, 0xFFFF, np.uint16)
        d1 = np.full(nq, 0xFFFF, np.uint16)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        rc = lib.uvo_hamming_knn2(self._h, _ptr(q), nq, _ptr(t), nt, _ptr(m), _ptr(idx0), _ptr(d0), _ptr(idx1), _ptr(d1))
        if rc:
            raise UvoError(rc, "uvo_hamming_knn2")
        return idx0, d0, idx1, d1

    def ratio_matching(self, q, t, ratio, mask=None):
        """Utils::ratioMatching (include/utils.h:81-111): (queryIdx, trainIdx, distance) of accepted matches."""
        if len(q) == 0 or len(t) == 0:
            return np.zeros((0, 3), np.int32)
        idx0, d0, idx1, d1 = self.knn2(q, t, mask)
        ok = (idx1 >= 0) & (d0.astype(np.float32).astype(np.float64) <= d1.astype(np.float32).astype(np.float64) * float(ratio))
        qi = np.nonzero(ok)[0].astype(np.int32)
        return np.stack([qi, idx0[qi], d0[qi].astype(np.int32)], 1)

    def distance_matrix(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros((len(q), len(t)), np.uint16)
        rc = lib.uvo_hamming_matrix(self._h, _ptr(q), len(q), _ptr(t), len(t), _ptr(out))
        if rc:
            raise UvoError(rc, "uvo_hamming_matrix")
        return out

    def distinctive_descriptors(self, desc_lists):
        """MapPoint::ComputeDistinctiveDescriptors for a batch of map points (src/MapPoint.cc:197-270).
        desc_lists: list of (n_i, 32) uint8 arrays.  Returns (best_idx, best_median) int32 arrays."""
        offs = np.zeros(len(desc_lists) + 1, np.int32)
        offs[1:] = np.cumsum([len(d) for d in desc_lists])
        allv = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in desc_lists if len(d)]
        allv = np.concatenate(allv) if allv else np.zeros((0, 32), np.uint8)
        idx = np.zeros(len(desc_lists), np.int32)
        med = np.zeros(len(desc_lists), np.int32)
        rc = lib.uvo_distinctive_descriptors(self._h, _ptr(allv) if len(allv) else None, _ptr(offs), len(desc_lists), _ptr(idx), _ptr(med))
        if rc:
            raise UvoError(rc, "uvo_distinctive_descriptors")
        return idx, med

    def knn2_batch_device(self, pairs, d_q, d_nq, q_stride, d_t, d_nt, t_stride, d_idx0, d_d0, d_idx1, d_d1):
        rc = lib.uvo_hamming_knn2_batch_device(self._h, pairs, d_q, d_nq, q_stride, d_t, d_nt, t_stride, d_idx0, d_d0, d_idx1, d_d1)
        if rc:
            raise UvoError(rc, "uvo_hamming_knn2_batch_device")

    def synchronize(self):
        rc = lib.uvo_matcher_synchronize(self._h)
        if rc:
            raise UvoError(rc, "uvo_matcher_synchronize")

    def wait_extractor(self, ex):
        rc = lib.uvo_matcher_wait_extractor(self._h, ex._h)
        if rc:
            raise UvoError(rc, "uvo_matcher_wait_extractor")

    def attach(self, ex):
        """Enqueue the matcher's work in the stream of the extractor's current lane from now on (None: back to its own stream)."""
        rc = lib.uvo_matcher_attach_extractor(self._h, ex._h if ex is not None else None)
        if rc:
            raise UvoError(rc, "uvo_matcher_attach_extractor")

    def release_to_extractor(self, ex):
        rc = lib.uvo_extractor_wait_matcher(ex._h, self._h)
        if rc:
            raise UvoError(rc, "uvo_extractor_wait_matcher")

    def profile(self, enable=True):
        lib.uvo_matcher_profile(self._h, 1 if enable else 0)

    def kernel_times(self):
        """{kernel: (summed ms, launches)}; the per-launch spread rows of the same report ("name:min" ...) are kept in self.last_spread
        as {kernel: {"min" | "p50" | "max" | "period_min" | "period_p50" | "period_max": ms}}."""
        names = ctypes.create_string_buffer(16384)
        ms = np.zeros(256, np.float32)
        launches = np.zeros(256, np.int32)
        n = ctypes.c_int()
        rc = lib.uvo_matcher_kernel_times(self._h, names, 16384, ms.ctypes.data, launches.ctypes.data, 256, ctypes.byref(n))
        if rc:
            raise UvoError(rc, "uvo_matcher_kernel_times")
        nm = names.value.decode().split("\n")[:n.value]
        self.last_spread = _spread_rows(nm, ms)
        return {nm[i]: (float(ms[i]), int(launches[i])) for i in range(n.value) if ":" not in nm[i]}

    def SearchByProjection(self, kp, desc, bounds, assigned, proj_x, proj_y, level, view_cos, in_view, mp_desc, scale_factors, th=1.0):
        """SearchByProjection(FrameKTL&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:49-125).

        kp/desc: frame keypoints (KEYPOINT_DTYPE) and descriptors; bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY);
        assigned: int32[n] in/out (-1 = F.mvpMapPoints[i] is NULL).  Returns nmatches.
        """
        kp = np.ascontiguousarray(kp, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        px, py = np.ascontiguousarray(proj_x, np.float32), np.ascontiguousarray(proj_y, np.float32)
        lv, vc = np.ascontiguousarray(level, np.int32), np.ascontiguousarray(view_cos, np.float32)
        iv, md = np.ascontiguousarray(in_view, np.uint8), np.ascontiguousarray(mp_desc, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        assert assigned.dtype == np.int32 and assigned.flags.c_contiguous
        nm = ctypes.c_int()
        rc = lib.uvo_search_by_projection(self._h, _ptr(kp), len(kp), _ptr(desc), int(bounds[0]), int(bounds[1]), int(bounds[2]), int(bounds[3]),
                                          _ptr(assigned), len(px), _ptr(px), _ptr(py), _ptr(lv), _ptr(vc), _ptr(iv), _ptr(md), _ptr(sf), len(sf),
                                          float(th), self.mfNNratio, ctypes.byref(nm))
        if rc:
            raise UvoError(rc, "uvo_search_by_projection")
        return nm.value


    # ---- the other search loops (generic engine + reference-named entry points) ----
    def match_windows(self, kp, desc, bounds, qx, qy, qr, qmin_level, qmax_level, qvalid, qdesc, rule, max_dist, *, blocked=None, qangle=None,
                      exclusive=True, check_orientation=False):
        """uvo_match_windows: candidates from 64x48 grid windows over kp; returns (match[nq], dist[nq], nmatches)."""
        kp = np.ascontiguousarray(kp, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        a = [np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr)]
        lo, hi = np.ascontiguousarray(qmin_level, np.int32), np.ascontiguousarray(qmax_level, np.int32)
        qv, qd = np.ascontiguousarray(qvalid, np.uint8), np.ascontiguousarray(qdesc, np.uint8)
        bl = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
        qa = None if qangle is None else np.ascontiguousarray(qangle, np.float32)
        nq = len(a[0])
        match, dist, nm = np.full(nq, -1, np.int32), np.full(nq, -1, np.int32), ctypes.c_int()
        r = MatchRule(rule, max_dist, self.mfNNratio, 1 if exclusive else 0, 1 if check_orientation else 0)
        rc = lib.uvo_match_windows(self._h, _ptr(kp), len(kp), _ptr(desc), _ptr(bl), int(bounds[0]), int(bounds[1]), int(bounds[2]), int(bounds[3]),
                                   nq, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(lo), _ptr(hi), _ptr(qv), _ptr(qd), _ptr(qa), ctypes.byref(r),
                                   _ptr(match), _ptr(dist), ctypes.byref(nm))
        if rc:
            raise UvoError(rc, "uvo_match_windows")
        return match, dist, nm.value

    def SearchByProjectionKF(self, kp, desc, bounds, assigned, u, v, level, valid, mp_desc, kf_angle, scale_factors, th, ORBdist):
        """SearchByProjection(FrameKTL& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1622-1746),
        from the projected (u, v, level) of the key frame's map points.  assigned in/out; returns nmatches."""
        kp = np.ascontiguousarray(kp, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        uu, vv = np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32)
        lv, va = np.ascontiguousarray(level, np.int32), np.ascontiguousarray(valid, np.uint8)
        md, ka = np.ascontiguousarray(mp_desc, np.uint8), np.ascontiguousarray(kf_angle, np.float32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        assert assigned.dtype == np.int32 and assigned.flags.c_contiguous
        nm = ctypes.c_int()
        rc = lib.uvo_search_by_projection_kf(self._h, _ptr(kp), len(kp), _ptr(desc), int(bounds[0]), int(bounds[1]), int(bounds[2]), int(bounds[3]),
                                             _ptr(assigned), len(uu), _ptr(uu), _ptr(vv), _ptr(lv), _ptr(va), _ptr(md), _ptr(ka), _ptr(sf), len(sf),
                                             float(th), int(ORBdist), 1 if self.mbCheckOrientation else 0, ctypes.byref(nm))
        if rc:
            raise UvoError(rc, "uvo_search_by_projection_kf")
        return nm.value

    def SearchByBoW(self, fv1, desc1, angle1, usable1, fv2, desc2, angle2, usable2=None, kf_kf=False):
        """SearchByBoW(KeyFrame*, FrameKTL&, ...) (:155-284) / SearchByBoW(KeyFrame*, KeyFrame*, ...) (:715-850, kf_kf=True).
        Returns (match12[n1], nmatches)."""
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        a1, a2 = np.ascontiguousarray(angle1, np.float32), np.ascontiguousarray(angle2, np.float32)
        u1 = np.ascontiguousarray(usable1, np.uint8)
        u2 = None if usable2 is None else np.ascontiguousarray(usable2, np.uint8)
        match = np.full(len(d1), -1, np.int32)
        nm = ctypes.c_int()
        rc = lib.uvo_search_by_bow(self._h, 1 if kf_kf else 0, ctypes.byref(fv1.c), len(d1), _ptr(d1), _ptr(a1), _ptr(u1), ctypes.byref(fv2.c), len(d2),
                                   _ptr(d2), _ptr(a2), _ptr(u2), self.mfNNratio, 1 if self.mbCheckOrientation else 0, _ptr(match), ctypes.byref(nm))
        if rc:
            raise UvoError(rc, "uvo_search_by_bow")
        return match, nm.value

    def SearchForTriangulation(self, fv1, kp1, desc1, has_mp1, fv2, kp2, desc2, has_mp2, F12, sigma2):
        """SearchForTriangulation(pKF1, pKF2, F12, ...) (:852-1014).  Returns (match12[n1], nmatches)."""
        kp1, kp2 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE), np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        h1, h2 = np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
        f, s2 = np.ascontiguousarray(F12, np.float32).reshape(9), np.ascontiguousarray(sigma2, np.float32)
        match = np.full(len(kp1), -1, np.int32)
        nm = ctypes.c_int()
        rc = lib.uvo_search_for_triangulation(self._h, ctypes.byref(fv1.c), _ptr(kp1), len(kp1), _ptr(d1), _ptr(h1), ctypes.byref(fv2.c), _ptr(kp2),
                                              len(kp2), _ptr(d2), _ptr(h2), _ptr(f), _ptr(s2), len(s2), 1 if self.mbCheckOrientation else 0,
                                              _ptr(match), ctypes.byref(nm))
        if rc:
            raise UvoError(rc, "uvo_search_for_triangulation")
        return match, nm.value

    def SearchForTriangulationBatch(self, fv1, kp1, desc1, has_mp1, pairs):
        """uvo_search_for_triangulation_batch: the distances + epipolar tests of every (pKF1, pKF2_k) pair of CreateNewMapPoints' loop
        (src/LocalMapping.cc:1058-1080) in one launch.  pairs = [(fv2, kp2, desc2, has_mp2, F12, sigma2), ...].  Follow with
        SearchForTriangulationNext(k, has_mp1 as it is then) per pair."""
        kp1 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE)
        d1, h1 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(has_mp1, np.uint8)
        keep = [kp1, d1, h1, fv1]
        arr = (TriangulationPairC * max(len(pairs), 1))()
        for k, (fv2, kp2, desc2, has_mp2, F12, sigma2) in enumerate(pairs):
            kp2 = np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
            d2, h2 = np.ascontiguousarray(desc2, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
            s2 = np.ascontiguousarray(sigma2, np.float32)
            keep += [fv2, kp2, d2, h2, s2]
            arr[k].fv2, arr[k].kp2, arr[k].n2, arr[k].desc2, arr[k].has_mp2 = ctypes.addressof(fv2.c), _ptr(kp2), len(kp2), _ptr(d2), _ptr(h2)
            arr[k].f12[:] = [float(x) for x in np.asarray(F12, np.float32).reshape(9)]
            arr[k].sigma2, arr[k].nlevels = _ptr(s2), len(s2)
        rc = lib.uvo_search_for_triangulation_batch(self._h, ctypes.byref(fv1.c), _ptr(kp1), len(kp1), _ptr(d1), _ptr(h1), len(pairs), arr)
        if rc:
            raise UvoError(rc, "uvo_search_for_triangulation_batch")
        self._tri_n1 = len(kp1)

    def SearchForTriangulationNext(self, pair, has_mp1_now):
        """uvo_search_for_triangulation_next: the acceptance loop of src/ORBmatcher.cc:886-984 for one pair of the batch, on the host.
        Returns (match12[n1], nmatches)."""
        h1 = np.ascontiguousarray(has_mp1_now, np.uint8)
        match = np.full(self._tri_n1, -1, np.int32)
        nm = ctypes.c_int()
        rc = lib.uvo_search_for_triangulation_next(self._h, int(pair), _ptr(h1), 1 if self.mbCheckOrientation else 0, _ptr(match), ctypes.byref(nm))
        if rc:
            raise UvoError(rc, "uvo_search_for_triangulation_next")
        return match, nm.value

    def FuseBatch(self, targets, xyz, normal, min_distance, max_distance, usable, mp_desc, th=3.0):
        """uvo_fuse_batch: projection tests + search core of Fuse (:1037-1101) for every (target key frame, map point) in one pass.
        targets = [(kp, desc, cam (CameraPose), scale_factors), ...]; min_distance / max_distance = the map points' mfMinDistance /
        mfMaxDistance (the invariance bounds x 0.8f / x 1.2f are formed here, as in project_points).  Returns
        (best_idx[n_targets][nmp], best_dist[n_targets][nmp])."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        mn_inv = (np.float32(0.8) * np.ascontiguousarray(min_distance, np.float32)).astype(np.float32)
        mx_inv = (np.float32(1.2) * np.ascontiguousarray(max_distance, np.float32)).astype(np.float32)
        us = None if usable is None else np.ascontiguousarray(usable, np.uint8)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        nmp = len(xyz)
        keep = []
        arr = (FuseTargetC * max(len(targets), 1))()
        for t, (kp, desc, cam, sf) in enumerate(targets):
            kp, desc, sf = np.ascontiguousarray(kp, KEYPOINT_DTYPE), np.ascontiguousarray(desc, np.uint8), np.ascontiguousarray(sf, np.float32)
            keep += [kp, desc, sf]
            arr[t].kp, arr[t].n, arr[t].desc = _ptr(kp), len(kp), _ptr(desc)
            arr[t].min_x, arr[t].min_y, arr[t].max_x, arr[t].max_y = int(cam.min_x), int(cam.min_y), int(cam.max_x), int(cam.max_y)
            arr[t].cam = cam
            arr[t].scale_factors, arr[t].nlevels = _ptr(sf), len(sf)
        bi, bd = np.full((len(targets), nmp), -1, np.int32), np.full((len(targets), nmp), -1, np.int32)
        rc = lib.uvo_fuse_batch(self._h, len(targets), arr, nmp, _ptr(xyz), _ptr(nrm), _ptr(mn_inv), _ptr(mx_inv), _ptr(us), _ptr(md), float(th), _ptr(bi), _ptr(bd))
        if rc:
            raise UvoError(rc, "uvo_fuse_batch")
        return bi, bd

    def project_points(self, mode, cam, xyz, normal, min_distance, max_distance, usable, scale_factors, scale_factor=1.2, viewing_cos_limit=0.5):
        """uvo_project_points: FrameKTL::isInFrustum (PROJECT_FRUSTUM), the projection prologue of SearchByProjection(F, pKF, ...)
        (PROJECT_KF_RELOC) or of Fuse (PROJECT_FUSE).  min_distance / max_distance are the map points' mfMinDistance / mfMaxDistance
        members; the invariance bounds (x 0.8f, x 1.2f: MapPoint::GetMin/MaxDistanceInvariance) are formed here in fp32.
        Returns (valid, u, v, level, view_cos)."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        nrm = None if normal is None else np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        mx = None if max_distance is None else np.ascontiguousarray(max_distance, np.float32)
        mn_inv = None if min_distance is None else (np.float32(0.8) * np.ascontiguousarray(min_distance, np.float32)).astype(np.float32)
        mx_inv = None if mx is None else (np.float32(1.2) * mx).astype(np.float32)
        us = None if usable is None else np.ascontiguousarray(usable, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        n = len(xyz)
        valid, u, v = np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.float32)
        level, vc = np.zeros(n, np.int32), np.zeros(n, np.float32)
        rc = lib.uvo_project_points(self._h, int(mode), ctypes.byref(cam), n, _ptr(xyz), _ptr(nrm), _ptr(mn_inv), _ptr(mx_inv), _ptr(mx), _ptr(us), _ptr(sf), len(sf),
                                    float(scale_factor), float(viewing_cos_limit), _ptr(valid), _ptr(u), _ptr(v), _ptr(level), _ptr(vc))
        if rc:
            raise UvoError(rc, "uvo_project_points")
        return valid, u, v, level, vc

    # ---- the four members of the reference class that nothing in the reference calls (src/ORBmatcher.cc:409-713, :1507-1620) ----
    def WindowSearch(self, kp1, desc1, has_mp1, kp2, desc2, bounds2, windowSize, minScaleLevel=-1, maxScaleLevel=0x7fffffff):
        """WindowSearch(F1, F2, windowSize, vpMapPointMatches2, minScaleLevel, maxScaleLevel) (:409-516).  has_mp1[i1] = F1's keypoint
        holds a good map point.  Returns (match21[n2] = index into F1 or -1, nmatches)."""
        kp1 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE)
        lv = kp1["octave"].astype(np.int32)
        valid = np.ascontiguousarray(has_mp1, np.uint8).copy()
        if minScaleLevel > 0:
            valid &= (lv >= minScaleLevel).astype(np.uint8)
        if maxScaleLevel < 0x7fffffff:
            valid &= (lv <= maxScaleLevel).astype(np.uint8)
        m12, _, nm = self.match_windows(kp2, desc2, bounds2, kp1["x"], kp1["y"], np.full(len(kp1), float(windowSize), np.float32), lv, lv, valid, desc1,
                                        RULE_BEST_RATIO_LEQ, 100, qangle=kp1["angle"], exclusive=True, check_orientation=self.mbCheckOrientation)
        m21 = np.full(len(kp2), -1, np.int32)
        sel = np.nonzero(m12 >= 0)[0]
        m21[m12[sel]] = sel
        return m21, nm

    def SearchByProjectionFrames(self, kp1, desc1, usable1, xyz1, cam2, kp2, desc2, assigned2, windowSize):
        """SearchByProjection(F1, F2, windowSize, vpMapPointMatches2) (:519-594).  usable1[i1]: F1's map point exists, is good and is not
        one of F2's already; cam2: F2's pose / intrinsics / bounds; assigned2 in/out (>= 0 = F2's keypoint holds a map point; new
        matches get the F1 index).  Returns nmatches."""
        kp1 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE)
        n1 = len(kp1)
        valid, u, v, _, _ = self.project_points(PROJECT_PIXEL, cam2, xyz1, None, None, None, usable1, np.ones(1, np.float32))
        lv = kp1["octave"].astype(np.int32)
        bounds = (int(cam2.min_x), int(cam2.min_y), int(cam2.max_x), int(cam2.max_y))
        m12, _, nm = self.match_windows(kp2, desc2, bounds, u, v, np.full(n1, float(windowSize), np.float32), lv, lv, valid, desc1, RULE_BEST_RATIO_LEQ, 100,
                                        blocked=(assigned2 >= 0).astype(np.uint8), exclusive=True, check_orientation=False)
        sel = np.nonzero(m12 >= 0)[0]
        assigned2[m12[sel]] = sel
        return nm

    def SearchForInitialization(self, kp1, desc1, kp2, desc2, bounds2, prev_matched, windowSize=10):
        """SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (:598-713).  prev_matched: float32 [n1][2], updated in
        place for the matched keypoints (:705-708).  Returns (vnMatches12[n1], nmatches)."""
        kp1, kp2 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE), np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
        n1 = len(kp1)
        assert prev_matched.dtype == np.float32 and prev_matched.shape == (n1, 2)
        lv = kp1["octave"].astype(np.int32)
        valid = (lv <= 0).astype(np.uint8)                                   # :621-622: level 0 only
        m12, _, nm = self.match_windows(kp2, desc2, bounds2, prev_matched[:, 0].copy(), prev_matched[:, 1].copy(), np.full(n1, float(windowSize), np.float32),
                                        lv, lv, valid, desc1, RULE_INIT_STEAL, 50, qangle=kp1["angle"], exclusive=False,
                                        check_orientation=self.mbCheckOrientation)
        sel = np.nonzero(m12 >= 0)[0]
        prev_matched[sel, 0], prev_matched[sel, 1] = kp2["x"][m12[sel]], kp2["y"][m12[sel]]
        return m12, nm

    def SearchByProjectionLast(self, kp, desc, assigned, cam, usable_last, xyz_last, octave_last, angle_last, desc_last, scale_factors, th):
        """SearchByProjection(CurrentFrame, LastFrame, th) (:1507-1620): LastFrame's map points projected with the current pose, window
        th * scale[octave] on levels [octave-1, octave+1], best <= TH_HIGH, first come first served, rotation histogram.  assigned
        in/out (new matches hold the LastFrame index).  Returns nmatches."""
        valid, u, v, _, _ = self.project_points(PROJECT_PIXEL_BOUNDED, cam, xyz_last, None, None, None, usable_last, scale_factors)
        bounds = (int(cam.min_x), int(cam.min_y), int(cam.max_x), int(cam.max_y))
        return self.SearchByProjectionKF(kp, desc, bounds, assigned, u, v, np.ascontiguousarray(octave_last, np.int32), valid, desc_last, angle_last,
                                         scale_factors, th, 100)

    def haloc_hash(self, proj, desc):
        """haloc::Hash::getHash (src/hash.cpp:57-85): proj [num_proj][>= n] float32, desc [n][32] -> hash [num_proj * 32]."""
        proj = np.ascontiguousarray(proj, np.float32)
        desc = np.ascontiguousarray(desc, np.uint8)
        out = np.zeros(proj.shape[0] * 32, np.float32)
        rc = lib.uvo_haloc_hash(self._h, _ptr(proj), proj.shape[0], proj.shape[1], _ptr(desc), len(desc), _ptr(out))
        if rc:
            raise UvoError(rc, "uvo_haloc_hash")
        return out

    def SearchPointsInFrustum(self, kp, desc, assigned, cam, xyz, normal, min_distance, max_distance, usable, mp_desc, scale_factors,
                              scale_factor=1.2, viewing_cos_limit=0.5, th=1.0, want_projections=False):
        """Tracking::SearchReferencePointsInFrustum (src/Tracking.cc:2176-2230) in one call: isInFrustum on every map point, then
        SearchByProjection on the ones in view.  assigned (int32[n], -1 = free) is updated in place.  Returns (n_matches, in_view) or
        (n_matches, in_view, u, v, level, view_cos)."""
        kp = np.ascontiguousarray(kp, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        assert assigned.dtype == np.int32 and assigned.flags.c_contiguous and len(assigned) == len(kp)
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        mx = np.ascontiguousarray(max_distance, np.float32)
        mn_inv = (np.float32(0.8) * np.ascontiguousarray(min_distance, np.float32)).astype(np.float32)
        mx_inv = (np.float32(1.2) * mx).astype(np.float32)
        us = None if usable is None else np.ascontiguousarray(usable, np.uint8)
        md, sf = np.ascontiguousarray(mp_desc, np.uint8), np.ascontiguousarray(scale_factors, np.float32)
        n = len(xyz)
        in_view = np.zeros(n, np.uint8)
        u = v = lv = vc = None
        if want_projections:
            u, v, lv, vc = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.float32)
        ntm, nm = ctypes.c_int(), ctypes.c_int()
        rc = lib.uvo_search_points_in_frustum(self._h, _ptr(kp), len(kp), _ptr(desc), _ptr(assigned), ctypes.addressof(cam), n, _ptr(xyz), _ptr(nrm),
                                              _ptr(mn_inv), _ptr(mx_inv), _ptr(mx), _ptr(us), _ptr(md), _ptr(sf), len(sf), float(scale_factor),
                                              float(viewing_cos_limit), float(th), self.mfNNratio, _ptr(in_view), _ptr(u), _ptr(v), _ptr(lv),
                                              _ptr(vc), ctypes.byref(ntm), ctypes.byref(nm))
        if rc:
            raise UvoError(rc, "uvo_search_points_in_frustum")
        assert ntm.value == int(in_view.sum())
        return (nm.value, in_view, u, v, lv, vc) if want_projections else (nm.value, in_view)

    @staticmethod
    def sim3_decompose(scw, cam):
        """Head of SearchByProjection(pKF, Scw, ...) / Fuse(pKF, Scw, ...) (:299-303): fills cam.rcw / tcw / ow from a 4x4 (or 3x4) Scw."""
        scw = np.ascontiguousarray(scw, np.float32)
        assert scw.ndim == 2 and scw.shape[0] >= 3 and scw.shape[1] == 4
        rc = lib.uvo_sim3_decompose(_ptr(scw), 4, ctypes.addressof(cam))
        if rc:
            raise UvoError(rc, "uvo_sim3_decompose")
        return cam

    @staticmethod
    def sim3_relative(s12, r12, t12):
        """SearchBySim3 :1284-1287: returns (sR12, sR21, t21)."""
        r12, t12 = np.ascontiguousarray(r12, np.float32).reshape(3, 3), np.ascontiguousarray(t12, np.float32).reshape(3)
        a, b, c = np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.zeros(3, np.float32)
        rc = lib.uvo_sim3_relative(float(s12), _ptr(r12), _ptr(t12), _ptr(a), _ptr(b), _ptr(c))
        if rc:
            raise UvoError(rc, "uvo_sim3_relative")
        return a, b, c

    def project_sim3(self, r_own, t_own, s_r, t, cam_other, xyz, min_distance, max_distance, usable, scale_factors):
        """Per-point prologue of one direction of SearchBySim3 (:1323-1359).  min_distance / max_distance = mfMinDistance / mfMaxDistance
        (the invariance bounds x 0.8f / x 1.2f are formed here).  Returns (valid, u, v, level)."""
        ro, to = np.ascontiguousarray(r_own, np.float32).reshape(9), np.ascontiguousarray(t_own, np.float32).reshape(3)
        sr, tt = np.ascontiguousarray(s_r, np.float32).reshape(9), np.ascontiguousarray(t, np.float32).reshape(3)
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        mn_inv = (np.float32(0.8) * np.ascontiguousarray(min_distance, np.float32)).astype(np.float32)
        mx_inv = (np.float32(1.2) * np.ascontiguousarray(max_distance, np.float32)).astype(np.float32)
        us = None if usable is None else np.ascontiguousarray(usable, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        n = len(xyz)
        valid, u, v, level = np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        rc = lib.uvo_project_sim3(self._h, _ptr(ro), _ptr(to), _ptr(sr), _ptr(tt), ctypes.addressof(cam_other), n, _ptr(xyz), _ptr(mn_inv), _ptr(mx_inv),
                                  _ptr(us), _ptr(sf), len(sf), _ptr(valid), _ptr(u), _ptr(v), _ptr(level))
        if rc:
            raise UvoError(rc, "uvo_project_sim3")
        return valid, u, v, level

    def SearchByProjectionSim3(self, kp, desc, bounds, matched, u, v, level, valid, mp_desc, scale_factors, th):
        """Search core of SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (:357-398).  matched (int32[n], >= 0 = taken) is
        updated in place; returns the number of new matches."""
        kp = np.ascontiguousarray(kp, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        assert matched.dtype == np.int32 and matched.flags.c_contiguous and len(matched) == len(kp)
        uu, vv = np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32)
        lv, va = np.ascontiguousarray(level, np.int32), np.ascontiguousarray(valid, np.uint8)
        md, sf = np.ascontiguousarray(mp_desc, np.uint8), np.ascontiguousarray(scale_factors, np.float32)
        nm = ctypes.c_int()
        rc = lib.uvo_search_by_projection_sim3(self._h, _ptr(kp), len(kp), _ptr(desc), int(bounds[0]), int(bounds[1]), int(bounds[2]), int(bounds[3]),
                                               _ptr(matched), len(uu), _ptr(uu), _ptr(vv), _ptr(lv), _ptr(va), _ptr(md), _ptr(sf), len(sf), int(th),
                                               ctypes.byref(nm))
        if rc:
            raise UvoError(rc, "uvo_search_by_projection_sim3")
        return nm.value

    def SearchBySim3(self, kp1, desc1, bounds1, kp2, desc2, bounds2, proj12, mp_desc1, proj21, mp_desc2, scale_factors1, scale_factors2, th):
        """SearchBySim3 (:1361-1504) after the projections: proj12 = (valid, u, v, level) of KF1's points in KF2, proj21 the reverse.
        Returns (match12[n1], n_found)."""
        kp1, kp2 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE), np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        b1, b2 = np.ascontiguousarray(bounds1, np.int32), np.ascontiguousarray(bounds2, np.int32)
        va12, u12, v12, l12 = (np.ascontiguousarray(proj12[0], np.uint8), np.ascontiguousarray(proj12[1], np.float32),
                               np.ascontiguousarray(proj12[2], np.float32), np.ascontiguousarray(proj12[3], np.int32))
        va21, u21, v21, l21 = (np.ascontiguousarray(proj21[0], np.uint8), np.ascontiguousarray(proj21[1], np.float32),
                               np.ascontiguousarray(proj21[2], np.float32), np.ascontiguousarray(proj21[3], np.int32))
        m1, m2 = np.ascontiguousarray(mp_desc1, np.uint8), np.ascontiguousarray(mp_desc2, np.uint8)
        s1, s2 = np.ascontiguousarray(scale_factors1, np.float32), np.ascontiguousarray(scale_factors2, np.float32)
        match12 = np.full(len(kp1), -1, np.int32)
        nf = ctypes.c_int()
        rc = lib.uvo_search_by_sim3(self._h, _ptr(kp1), len(kp1), _ptr(d1), _ptr(b1), _ptr(kp2), len(kp2), _ptr(d2), _ptr(b2), _ptr(u12), _ptr(v12),
                                    _ptr(l12), _ptr(va12), _ptr(m1), _ptr(u21), _ptr(v21), _ptr(l21), _ptr(va21), _ptr(m2), _ptr(s1), len(s1), _ptr(s2),
                                    len(s2), float(th), _ptr(match12), ctypes.byref(nf))
        if rc:
            raise UvoError(rc, "uvo_search_by_sim3")
        return match12, nf.value

    def FuseSearch(self, kp, desc, bounds, u, v, level, valid, mp_desc, scale_factors, th=3.0):
        """Search core of Fuse (:1077-1101): (best_idx[nmp], best_dist[nmp]), -1 where nothing within TH_LOW."""
        kp = np.ascontiguousarray(kp, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        uu, vv = np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32)
        lv, va = np.ascontiguousarray(level, np.int32), np.ascontiguousarray(valid, np.uint8)
        md, sf = np.ascontiguousarray(mp_desc, np.uint8), np.ascontiguousarray(scale_factors, np.float32)
        bi, bd = np.full(len(uu), -1, np.int32), np.full(len(uu), -1, np.int32)
        rc = lib.uvo_fuse(self._h, _ptr(kp), len(kp), _ptr(desc), int(bounds[0]), int(bounds[1]), int(bounds[2]), int(bounds[3]), len(uu), _ptr(uu),
                          _ptr(vv), _ptr(lv), _ptr(va), _ptr(md), _ptr(sf), len(sf), float(th), _ptr(bi), _ptr(bd))
        if rc:
            raise UvoError(rc, "uvo_fuse")
        return bi, bd


class VocabularyDesc(ctypes.Structure):
    """uvo_vocabulary_desc."""
    _fields_ = [("n_nodes", ctypes.c_int32), ("child_start", ctypes.c_void_p), ("children", ctypes.c_void_p), ("descriptor", ctypes.c_void_p),
                ("word_id", ctypes.c_void_p), ("weight", ctypes.c_void_p), ("L", ctypes.c_int32), ("weighting", ctypes.c_int32),
                ("normalize", ctypes.c_int32), ("device", ctypes.c_int32)]


class ORBVocabulary:
    """DBoW2 ORBVocabulary stand-in holding the tree on the device: transform() = TemplatedVocabulary::transform
    (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1125-1258).  Tree given flat: children of node i =
    children[child_start[i]:child_start[i+1]] (node 0 = root), descriptor[n_nodes][32], word_id[n_nodes], weight[n_nodes]."""
    TF_IDF, TF, IDF, BINARY = range(4)

    def __init__(self, child_start, children, descriptor, word_id, weight, L, weighting=0, normalize=1, device=0):
        self._k = [np.ascontiguousarray(child_start, np.int32), np.ascontiguousarray(children, np.int32), np.ascontiguousarray(descriptor, np.uint8),
                   np.ascontiguousarray(word_id, np.int32), np.ascontiguousarray(weight, np.float64)]
        d = VocabularyDesc(len(self._k[0]) - 1, _ptr(self._k[0]), _ptr(self._k[1]), _ptr(self._k[2]), _ptr(self._k[3]), _ptr(self._k[4]), int(L),
                           int(weighting), int(normalize), int(device))
        self._h = ctypes.c_void_p()
        rc = lib.uvo_vocabulary_create(ctypes.byref(d), ctypes.byref(self._h))
        if rc:
            raise UvoError(rc, "uvo_vocabulary_create")

    def close(self):
        if self._h:
            lib.uvo_vocabulary_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform(self, desc, levelsup=4):
        """Returns (word_id[n], weight[n], node_id[n], BowVector as (ids, values), FeatureVector)."""
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        wid, nid, ww = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float64)
        bid, bval = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        fnode, fstart, ffeat = np.zeros(max(n, 1), np.uint32), np.zeros(n + 2, np.int32), np.zeros(max(n, 1), np.int32)
        nb, nf = ctypes.c_int(), ctypes.c_int()
        rc = lib.uvo_bow_transform(self._h, _ptr(desc), n, int(levelsup), _ptr(wid), _ptr(ww), _ptr(nid), _ptr(bid), _ptr(bval), len(bid), ctypes.byref(nb),
                                   _ptr(fnode), _ptr(fstart), _ptr(ffeat), len(fnode), ctypes.byref(nf))
        if rc:
            raise UvoError(rc, "uvo_bow_transform")
        groups = {int(fnode[j]): [int(x) for x in ffeat[fstart[j]:fstart[j + 1]]] for j in range(nf.value)}
        return wid, ww, nid, (bid[:nb.value].copy(), bval[:nb.value].copy()), FeatureVector(groups)


class CameraModel(ctypes.Structure):
    """uvo_camera_model: mK and mDistCoef of the reference's Tracking object (pin-hole: k1 k2 p1 p2 [k3 ..]; fisheye: 4 coefficients)."""
    _fields_ = [("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float), ("dist", ctypes.c_float * 8),
                ("n_dist", ctypes.c_int32), ("fisheye", ctypes.c_int32)]

    @classmethod
    def make(cls, fx, fy, cx, cy, dist, fisheye=False):
        c = cls(fx, fy, cx, cy)
        for i, v in enumerate(dist):
            c.dist[i] = v
        c.n_dist, c.fisheye = len(dist), 1 if fisheye else 0
        return c


class KltCfg(ctypes.Structure):
    """uvo_klt_cfg."""
    _fields_ = [("max_width", ctypes.c_int32), ("max_height", ctypes.c_int32), ("max_level", ctypes.c_int32), ("win_width", ctypes.c_int32),
                ("win_height", ctypes.c_int32), ("max_points", ctypes.c_int32), ("slots", ctypes.c_int32), ("device", ctypes.c_int32)]


class KLT:
    """cv::buildOpticalFlowPyramid (src/FrameKTL.cc:76) + cv::calcOpticalFlowPyrLK as called at src/Tracking.cc:1046-1047
    (USE_INITIAL_FLOW + LK_GET_MIN_EIGENVALS, 30 iterations / eps 0.01)."""

    def __init__(self, max_width, max_height, win=(21, 21), max_level=5, max_points=4096, slots=2, device=0):
        cfg = KltCfg(max_width, max_height, max_level, win[0], win[1], max_points, slots, device)
        self._h = ctypes.c_void_p()
        self.max_level = max_level
        rc = lib.uvo_klt_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc:
            raise UvoError(rc, "uvo_klt_create")

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.uvo_klt_destroy(self._h)
            self._h = None

    __del__ = close

    def build_pyramid(self, slot, img):
        img = np.ascontiguousarray(img, np.uint8)
        n = ctypes.c_int()
        rc = lib.uvo_klt_build_pyramid(self._h, slot, _ptr(img), img.shape[1], img.shape[0], img.strides[0], ctypes.byref(n))
        if rc:
            raise UvoError(rc, "uvo_klt_build_pyramid")
        return n.value

    def build_pyramid_from(self, slot, extractor):
        """Pyramid of the extractor's last clahe() result, taken from HBM (no upload)."""
        n = ctypes.c_int()
        rc = lib.uvo_klt_build_pyramid_from_extractor(self._h, slot, extractor._h, ctypes.byref(n))
        if rc:
            raise UvoError(rc, "uvo_klt_build_pyramid_from_extractor")
        return n.value

    def read_level(self, slot, level):
        w, h = ctypes.c_int(), ctypes.c_int()
        rc = lib.uvo_klt_read_level(self._h, slot, level, None, None, ctypes.byref(w), ctypes.byref(h))
        if rc:
            raise UvoError(rc, "uvo_klt_read_level")
        img, der = np.zeros((h.value, w.value), np.uint8), np.zeros((h.value, w.value, 2), np.int16)
        rc = lib.uvo_klt_read_level(self._h, slot, level, _ptr(img), _ptr(der), ctypes.byref(w), ctypes.byref(h))
        if rc:
            raise UvoError(rc, "uvo_klt_read_level")
        return img, der

    def track(self, prev_slot, next_slot, prev_pts, next_pts0=None, max_count=30, epsilon=0.01, min_eig_threshold=1e-4):
        """Returns (next_pts, status, err)."""
        p0 = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        p1 = p0.copy() if next_pts0 is None else np.ascontiguousarray(next_pts0, np.float32).reshape(-1, 2).copy()
        st, er = np.zeros(len(p0), np.uint8), np.zeros(len(p0), np.float32)
        rc = lib.uvo_klt_track(self._h, prev_slot, next_slot, _ptr(p0), _ptr(p1), len(p0), self.max_level, int(max_count), float(epsilon),
                               float(min_eig_threshold), _ptr(st), _ptr(er))
        if rc:
            raise UvoError(rc, "uvo_klt_track")
        return p1, st, er

    def undistort(self, cam, pts):
        """Tracking::undistort_point (src/Tracking.cc:1265-1283) for an (n, 2) float32 array; cam: CameraModel."""
        p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        out = np.zeros_like(p)
        rc = lib.uvo_undistort_points(self._h, ctypes.byref(cam), _ptr(p), len(p), _ptr(out))
        if rc:
            raise UvoError(rc, "uvo_undistort_points")
        return out

    def track_undistorted(self, prev_slot, next_slot, prev_pts, cam, next_pts0=None, max_count=30, epsilon=0.01, min_eig_threshold=1e-4):
        """uvo_klt_track_undistorted: the LK step + undistort_point of both point sets in one call.
        Returns (next, status, err, prev_un, next_un)."""
        a = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        b = a.copy() if next_pts0 is None else np.ascontiguousarray(next_pts0, np.float32).reshape(-1, 2).copy()
        n = len(a)
        st, er = np.zeros(n, np.uint8), np.zeros(n, np.float32)
        pu, nu = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
        rc = lib.uvo_klt_track_undistorted(self._h, prev_slot, next_slot, _ptr(a), _ptr(b), n, self.max_level, int(max_count), float(epsilon), float(min_eig_threshold),
                                           ctypes.byref(cam), _ptr(st), _ptr(er), _ptr(pu), _ptr(nu))
        if rc:
            raise UvoError(rc, "uvo_klt_track_undistorted")
        return b, st, er, pu, nu


def DescriptorDistance(a, b):
    """ORBmatcher::DescriptorDistance for one pair (src/ORBmatcher.cc:1794-1810); host convenience for scripts.
    Bulk distances go through ORBmatcher.knn2 / distance_matrix on the GPU."""
    return int(np.unpackbits(np.bitwise_xor(np.asarray(a, np.uint8), np.asarray(b, np.uint8))).sum())
