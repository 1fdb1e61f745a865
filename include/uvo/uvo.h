/*
 * uvo.h -- C ABI of the MI355X-native ORB feature front-end (extract + match).
 *
 * This is the drop-in boundary for the hot path of chintha/U-VIP-SLAM.  The reference has no FFI layer:
 * the path is entered through two concrete C++ classes.  Each entry point below names the reference
 * interface it replaces (file:line relative to the reference tree); include/uvo/compat/ holds C++
 * adaptors with the reference's own class/method signatures on top of this ABI, and INTEGRATION.md shows
 * the binding a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (UVO_OK) or a negative UVO_E_* code; nothing throws across the ABI.
 *   - plain pointers and sizes only; "host" pointers are ordinary CPU memory, "device" pointers are HBM
 *     addresses on the handle's GPU (hipMalloc / torch.cuda tensors).
 *   - a handle owns all device scratch (sized at create time), one HIP stream and its pinned staging;
 *     one handle = one in-flight call; use one handle per GPU / per host thread.
 *   - there is NO CPU fallback: if no gfx950 device is usable, create() fails with UVO_E_NODEVICE.
 */
#ifndef UVO_UVO_H_
#define UVO_UVO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVO_OK 0
#define UVO_E_BADARG (-1)      /* null pointer, non-positive size, image larger than the handle was sized for */
#define UVO_E_NODEVICE (-2)    /* no usable HIP device / kernels not loadable */
#define UVO_E_HIP (-3)         /* a HIP runtime call failed; see uvo_last_error() */
#define UVO_E_CAPACITY (-4)    /* caller's output capacity too small (n_out still reports what was needed) */
#define UVO_E_UNSUPPORTED (-5) /* geometry outside the supported envelope (see uvo_extractor_create) */
#define UVO_E_NOMEM (-6)

/* Layout-identical to cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id. */
typedef struct uvo_keypoint {
  float x, y;
  float size;
  float angle;
  float response;
  int32_t octave;
  int32_t class_id;
} uvo_keypoint;

/* ------------------------------------------------------------------------------------------------
 * Extractor -- replaces USLAM::ORBextractor (include/ORBextractor.h:47-95, src/ORBextractor.cc).
 * ---------------------------------------------------------------------------------------------- */
typedef struct uvo_extractor uvo_extractor;

typedef struct uvo_extractor_cfg {
  /* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, scoreType, fastTh): include/ORBextractor.h:51 */
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels;
  int32_t score_type; /* accepted and ignored, exactly like the live reference path */
  int32_t fast_th;    /* >= 0; values above 255 act as 255 (cv::FAST clamps its threshold the same way) */
  /* sizing of the device scratch owned by the handle */
  int32_t max_width, max_height; /* largest frame; every pyramid level must keep >= 56 px per side */
  int32_t max_batch;             /* frames per uvo_extract_batch* call */
  int32_t max_input_keypoints;   /* per frame: caller keypoints passed through level 0 (top-up mode) */
  int32_t device;                /* HIP device ordinal */
} uvo_extractor_cfg;

/*
 * The supported envelope.  The reference has none of these limits (it allocates as it goes); here they bound the kernels' LDS and
 * register budgets, and a configuration or image outside them is refused with UVO_E_UNSUPPORTED -- by uvo_extractor_create() for
 * max_width x max_height, and by every extract call for the size it is given -- never computed approximately:
 *   - every pyramid level must measure 56 .. 4096 px on each side (level l is round(size * invScaleFactor^l),
 *     src/ORBextractor.cc:966-969; below 56 px the 30-px FAST cell grid of :755-770 has no cell);
 *   - a FAST cell ROI (wCell + 6 x hCell + 6, :773-790) may not exceed 66 x 66 px (true whenever the detection window of the
 *     level is at least 30 px on each side);
 *   - a level's feature quota mnFeaturesPerLevel[l] (:472-485) may not exceed 2000 (with the reference's 1.2 / 8 levels that is
 *     nfeatures <= ~9200);
 *   - the number of quad-tree roots of a level, round(window width / window height) (:1010), must be 1 .. 64 (`nIni = 0`, an image
 *     more than twice as high as wide, divides by zero in the reference too);
 *   - at most 2^24 FAST cells per frame; nlevels <= 16.
 * uvo_sharder additionally needs uvo_extractor_max_keypoints() <= 65535 when it matches (train indices are packed in 16 bits).
 */
int uvo_extractor_create(const uvo_extractor_cfg* cfg, uvo_extractor** out);
void uvo_extractor_destroy(uvo_extractor* h);

/* ORBextractor::GetLevels / GetScaleFactor: include/ORBextractor.h:60-64 */
/* Largest number of keypoints one frame can return with this handle (sum over levels of max(quota, 4 * nIni) + 4, plus
 * max_input_keypoints): the output capacity that can never overflow.  Negative = error code. */
int uvo_extractor_max_keypoints(const uvo_extractor* h);
int uvo_extractor_levels(const uvo_extractor* h);
float uvo_extractor_scale_factor(const uvo_extractor* h);
/* constructor tables (mvScaleFactor, mvInvScaleFactor, mnFeaturesPerLevel, umax[16]): src/ORBextractor.cc:463-511 */
int uvo_extractor_tables(const uvo_extractor* h, float* scale, float* inv_scale, int32_t* quota, int32_t* umax16);

/*
 * ORBextractor::operator()(image, mask, keypoints, descriptors, grid_2d, min_px_dist, FullDetect,
 * num_featsneeded): include/ORBextractor.h:56-58, src/ORBextractor.cc:849-961; call site src/Tracking.cc:946.
 *   img/width/height/stride : CV_8UC1 host image
 *   in_kp/n_in              : caller keypoints (reference: `keypoints` on entry); passed through level 0
 *                             with recomputed angle when full_detect == 0, dropped when full_detect != 0
 *   grid2d                  : Eigen::MatrixXi::data() -- column-major int32, grid_rows x grid_cols,
 *                             read and MUTATED when full_detect == 0 (may be NULL when full_detect != 0)
 *   out_kp/out_desc/cap     : caller buffers for up to `cap` keypoints (28 B) and descriptors (32 B)
 *   n_out                   : number of keypoints produced (reference: keypoints.size() on return)
 * Keypoints come out level-major in the reference's list order; nothing is truncated to nfeatures.
 */
int uvo_extract(uvo_extractor* h, const uint8_t* img, int width, int height, ptrdiff_t stride, const uvo_keypoint* in_kp, int n_in,
                int32_t* grid2d, int grid_rows, int grid_cols, int min_px_dist, int full_detect, int num_feats_needed,
                uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int* n_out);

/*
 * The top-up call together with the caller's loop in front of it (src/Tracking.cc:896-946): the occupancy grid is built on the
 * device from the tracked keypoints -- grid_2d((int)(pt.y / min_px_dist), (int)(pt.x / min_px_dist))++ on a zero matrix of
 * (height / min_px_dist + 2) x (width / min_px_dist + 2) -- and the extraction runs with FullDetect = false on it, with an EMPTY
 * keypoint vector as at :946 (`pts0_ext`): in_kp are the tracked keypoints `pts0`, which only fill the grid (they must lie inside the
 * image); out_kp / out_desc hold the NEW keypoints and descriptors alone (`pts0_ext`, `New_Descriptors`), which the caller appends
 * to its own (:948-959).  grid2d_out (optional): the grid after the call, column-major, for callers that keep it.  img == NULL: the
 * result of the last uvo_clahe(), as in uvo_extract.
 */
int uvo_extract_tracked(uvo_extractor* h, const uint8_t* img, int width, int height, ptrdiff_t stride, const uvo_keypoint* in_kp, int n_in,
                        int min_px_dist, int num_feats_needed, uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int* n_out, int32_t* grid2d_out);

/*
 * Batched form of the same call (B independent frames, one launch per stage).  Host buffers:
 *   imgs           : frame b at imgs + b*frame_stride, rows `stride` bytes apart
 *   in_kp / n_in   : [B][max_input_keypoints] / [B]   (NULL / NULL when there are none)
 *   grid2d         : [B][grid_rows*grid_cols] column-major each (NULL when full_detect != 0)
 *   num_feats_needed : [B] (ignored when full_detect != 0; may be NULL then)
 *   out_kp/out_desc: [B][cap] / [B][cap][32];  n_out: [B]
 */
int uvo_extract_batch(uvo_extractor* h, int batch, const uint8_t* imgs, int width, int height, ptrdiff_t stride, ptrdiff_t frame_stride,
                      const uvo_keypoint* in_kp, const int32_t* n_in, int32_t* grid2d, int grid_rows, int grid_cols, int min_px_dist,
                      int full_detect, const int32_t* num_feats_needed, uvo_keypoint* out_kp, uint8_t* out_desc, int cap,
                      int32_t* n_out);

/*
 * Asynchronous host-buffer form (FullDetect): uvo_extract_batch_submit() enqueues the upload of the frames, the extraction and
 * the download of the results on the next pipeline lane and returns at once with a ticket (the lane); uvo_extract_batch_wait()
 * blocks until that lane has finished.  With uvo_extractor_set_pipeline(h, 2) and page-locked buffers (uvo_host_alloc) the
 * PCIe transfers of one batch overlap the kernels of the other: submit(k+1), wait(k), consume k, ...
 *   imgs, out_kp, out_desc, n_out must stay valid and untouched until the matching wait returns; a lane must be waited for
 *   before it is submitted to again (with depth d: at most d batches in flight).
 *   out_kp / out_desc: [B][cap] / [B][cap][32]; records past n_out[b] are unspecified.  cap must be >= uvo_extractor_max_keypoints().
 */
int uvo_host_alloc(void** ptr, size_t bytes);
int uvo_host_free(void* ptr);
int uvo_extract_batch_submit(uvo_extractor* h, int batch, const uint8_t* imgs, int width, int height, ptrdiff_t stride, ptrdiff_t frame_stride,
                             uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int32_t* n_out, int* ticket);
int uvo_extract_batch_wait(uvo_extractor* h, int ticket);
/*
 * Placement of the host side next to a GPU (N ranks on a two-socket host: every rank's frames leave over the PCIe link of ITS GPU).
 * uvo_host_bind_near_device() binds the CALLING THREAD to the CPUs local to `device` (Linux: /sys/bus/pci/devices/<bdf>/local_cpulist,
 * intersected with the CPUs the process may use) and reports the device's NUMA node (-1: unknown); memory the thread touches first
 * afterwards -- its frames, its slice of a shared gather region, before uvo_host_register() -- then lands on that node.  Returns 1 when the
 * thread was bound, 0 when there was nothing to do (no sysfs entry, no usable CPU in the list: the thread stays where it was), < 0 on error.
 * The sharder's per-shard threads bind themselves this way (environment UVO_NUMA_BIND=0 turns every binding of this library into a no-op:
 * for hosts that place their processes themselves).  uvo_host_bind_to_cpulist_file() is the same for an explicit cpulist file.
 */
int uvo_host_bind_near_device(int device, int32_t* numa_node);
int uvo_host_bind_to_cpulist_file(const char* cpulist_path);
/* Page-lock memory the caller already owns (e.g. a shared mapping that several processes gather into); uvo_host_alloc()'s sibling. */
int uvo_host_register(void* ptr, size_t bytes);
int uvo_host_unregister(void* ptr);

/* ------------------------------------------------------------------------------------------------
 * Sharder -- one job of `total` frames over N GPUs of one node, no collective.
 * The reference has one extractor call per frame on one thread (src/Tracking.cc:946); an offline
 * mapping run (BASELINE.json configs[3]) is that call over a whole sequence.  Frames are independent:
 * shard i owns one contiguous block, runs it on its own device from its own host thread in chunks
 * (two in flight: the upload of one under the kernels of the other), and every chunk's results are
 * copied straight to element frame * cap of ONE set of caller arrays -- the gather is those copies.
 * With match = 1 pair p = (frame p, frame p + 1) is matched (all-pairs knn-2, uvo_hamming_knn2) by the
 * shard that owns frame p; the last pair of a chunk needs the next frame's descriptors, so a chunk
 * extracts one halo frame beyond its own (at a shard's end: the neighbouring shard's first frame,
 * recomputed here, bit-identical to the neighbour's own copy) rather than exchanging descriptors.
 * ---------------------------------------------------------------------------------------------- */
#define UVO_SHARD_MAX 64
#define UVO_SHARD_REMOTE (-1) /* devices[i]: shard i is run by another process (same plan, same offsets) */
typedef struct uvo_sharder uvo_sharder;
typedef struct uvo_sharder_cfg {
  uvo_extractor_cfg extractor;    /* device / max_batch / max_input_keypoints are set per shard by the sharder */
  int32_t n_shards;
  int32_t devices[UVO_SHARD_MAX]; /* HIP device ordinal of shard i (ordinals may repeat), or UVO_SHARD_REMOTE */
  int32_t chunk_frames;           /* frames per chunk (per device launch sequence); device scratch is sized for chunk_frames + 1 */
  int32_t match;                  /* 1: also the knn-2 rows of consecutive frames */
} uvo_sharder_cfg;
/* The block of shard `shard` and what it matches -- pure arithmetic, no device needed. */
typedef struct uvo_shard_plan {
  int32_t first_frame, n_frames; /* frames [first_frame, first_frame + n_frames): results at element first_frame * cap of the outputs */
  int32_t first_pair, n_pairs;   /* pairs p = (p, p + 1), p in [first_pair, first_pair + n_pairs): rows at element p * cap */
  int32_t halo_frame;            /* the neighbouring shard's first frame, extracted here too for the block's last pair; -1 = none */
  int32_t n_chunks;
} uvo_shard_plan;
int uvo_shard_plan_make(int total_frames, int n_shards, int shard, int chunk_frames, uvo_shard_plan* out);
int uvo_sharder_create(const uvo_sharder_cfg* cfg, uvo_sharder** out);
void uvo_sharder_destroy(uvo_sharder* s);
int uvo_sharder_max_keypoints(const uvo_sharder* s); /* smallest `cap` uvo_sharder_run accepts */
/*
 * Run the local shards of a job.  uvo_sharder_submit() queues the job for the shard threads and returns a ticket at once,
 * uvo_sharder_wait() blocks until the job's results are in the output arrays; uvo_sharder_run() is the two together.  Jobs run in
 * submission order and the lanes are not drained between them: with a second job submitted before the first is waited for, the
 * uploads of one run under the kernels of the other (a stream of jobs moves at the steady-state rate of the pipeline lanes).
 *   imgs, n_imgs     : n_imgs host frames, frame g at imgs + (g - imgs_first_frame) * frame_stride (a process that owns only some
 *                      shards need only hold their frames and each block's halo frame: UVO_E_BADARG when a local shard's frames or
 *                      its halo frame lie outside [imgs_first_frame, imgs_first_frame + n_imgs)); page-locked memory makes the
 *                      uploads asynchronous
 *   out_kp / out_desc / n_out : [total][cap] / [total][cap][32] / [total]   (cap >= uvo_sharder_max_keypoints())
 *   idx0 / d0 / idx1 / d1     : [total - 1][cap] knn-2 rows of pair p (row q = keypoint q of frame p; as uvo_hamming_knn2), or all NULL
 * Only the elements of the local shards' frames / pairs are written.  All buffers of a job must stay valid and untouched until its
 * wait returns; jobs in flight at the same time need their own output arrays.
 */
int uvo_sharder_submit(uvo_sharder* s, const uint8_t* imgs, int n_imgs, int imgs_first_frame, int total_frames, int width, int height,
                       ptrdiff_t stride, ptrdiff_t frame_stride, uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int32_t* n_out, int32_t* idx0, uint16_t* d0,
                       int32_t* idx1, uint16_t* d1, int* ticket);
int uvo_sharder_wait(uvo_sharder* s, int ticket);
int uvo_sharder_run(uvo_sharder* s, const uint8_t* imgs, int n_imgs, int imgs_first_frame, int total_frames, int width, int height,
                    ptrdiff_t stride, ptrdiff_t frame_stride, uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int32_t* n_out, int32_t* idx0, uint16_t* d0,
                    int32_t* idx1, uint16_t* d1);

/*
 * HBM-resident form: every pointer is a device pointer on the handle's GPU and the call only enqueues work
 * on the handle's stream (uvo_extractor_synchronize() waits for it).  Same argument meaning as above.
 * d_n_out[b] may exceed `cap`; only the first `cap` records of a frame are written in that case.
 * A pipeline lane keeps at most two batches outstanding: the call first waits (on the host) for the lane's last but one batch, so a
 * caller that enqueues in a loop runs two to four batches ahead of the device and no further.
 * d_imgs must stay valid AND UNCHANGED until the batch has completed: with dword-aligned rows (d_imgs, stride, frame_stride multiples of
 * 4, width a multiple of 4) and no caller keypoints, level 0 of the pyramid is the caller's image itself, read in place by every stage
 * (UVO_TUNE_LEVEL0_INPLACE; cv::copyMakeBorder of src/ORBextractor.cc:996 is never materialised).  Results do not depend on it.
 */
int uvo_extract_batch_device(uvo_extractor* h, int batch, const uint8_t* d_imgs, int width, int height, ptrdiff_t stride,
                             ptrdiff_t frame_stride, const uvo_keypoint* d_in_kp, const int32_t* d_n_in, int32_t* d_grid2d, int grid_rows,
                             int grid_cols, int min_px_dist, int full_detect, const int32_t* d_num_feats_needed, uvo_keypoint* d_out_kp,
                             uint8_t* d_out_desc, int cap, int32_t* d_n_out);
int uvo_extractor_synchronize(uvo_extractor* h);
/*
 * Pipeline depth of the HBM-resident form (default 1).  With depth 2 the handle owns two scratch sets and two
 * streams and consecutive uvo_extract_batch_device() calls alternate between them, so the latency-bound stages of one
 * batch overlap with the streaming stages of the next.  The caller must then give consecutive calls different output
 * buffers.  uvo_matcher_wait_extractor() / uvo_extractor_wait_matcher() refer to the most recently used stream.
 */
int uvo_extractor_set_pipeline(uvo_extractor* h, int depth);
/*
 * Launch-shape knobs of one handle (speed only -- results never depend on them; the parity tests force each shape).
 *   UVO_TUNE_OCT_WIDE_MAX : batches with at most this many (frame, level) quad-tree problems run DistributeOctTree
 *                           (src/ORBextractor.cc:1006-1230) as 1024-thread workgroups, larger ones as 256-thread workgroups
 *                           (default 0 = always the 256-thread form, which shares its launch with the blur; a large value = always the 1024-thread form).
 */
#define UVO_TUNE_OCT_WIDE_MAX 1
/*
 *   UVO_TUNE_FAST_MODE    : how the per-cell threshold fallback of src/ORBextractor.cc:792-799 (`FAST(cell, fastTh)`, and
 *                           `FAST(cell, 7)` when that finds nothing) is computed when fastTh > 7.
 *                           UVO_FAST_MODE_TWO_PASS    every level streams once at fastTh; the cells left without a keypoint are
 *                                                     redone at the literal 7 by a sparse per-cell kernel;
 *                           UVO_FAST_MODE_SINGLE_PASS every level streams once at 7 and the per-cell vote picks the class;
 *                           UVO_FAST_MODE_ADAPTIVE    (default) per level and pipeline lane, whichever was cheaper for the previous
 *                                                     batch's share of fall-back cells (decided on the device, no host round trip).
 *                           The keypoints are the same in every mode.
 */
#define UVO_TUNE_FAST_MODE 2
#define UVO_FAST_MODE_ADAPTIVE 0
#define UVO_FAST_MODE_TWO_PASS 1
#define UVO_FAST_MODE_SINGLE_PASS 2
/*
 *   UVO_TUNE_BLUR_ROUNDING : the ONE knob that changes results -- which OpenCV build's GaussianBlur (src/ORBextractor.cc:942) is
 *                           reproduced where the two differ: an exact .5 in the column pass of the 7 x 7 blur (about one pixel in
 *                           65 536, by one grey level; descriptors follow).
 *                           UVO_BLUR_ROUNDING_SSE2   (default) x86-64 builds -- the reference is an x86-64 ROS program: the vector body of
 *                                                     SymmColumnVec_32s8u (image columns 0 .. (w & ~3) - 1) converts the exact fp32 sum with
 *                                                     cvtps2dq, half to EVEN; only the last w % 4 columns take the scalar loop, half up;
 *                           UVO_BLUR_ROUNDING_SCALAR  the generic C++ column filter: (sum + 2^15) >> 16, half up, on every column (builds
 *                                                     without SIMD).
 *                           tools/pin/ dumps the evidence that decides it for a given OpenCV binary (the gauss_<k> cases).
 */
#define UVO_TUNE_LEVEL0_INPLACE 11 /* 1 (default): level 0 is read from the caller's image in place whenever it can be (dword-aligned rows, width a
                                     multiple of 4, no caller keypoints): cv::copyMakeBorder of src/ORBextractor.cc:996 is never materialised, the
                                     blur reflects the border it needs on the fly; 0: always copy into a padded plane first */
#define UVO_TUNE_BLUR_ROUNDING 9
#define UVO_BLUR_ROUNDING_SCALAR 0
#define UVO_BLUR_ROUNDING_SSE2 1
int uvo_extractor_tune(uvo_extractor* h, int knob, int value);
/*
 * State of the adaptive FAST mode after the most recent batch (waits for it): per level the threshold the NEXT batch on that
 * pipeline lane streams at (fastTh = two-pass form, <= 7 = single pass), the fall-back cells the last batch counted (cells without a
 * keypoint at fastTh, summed over its frames) and the level's cells per frame.  Arrays of uvo_extractor_levels() entries; any may be NULL.
 */
int uvo_extractor_fast_state(uvo_extractor* h, int32_t* pass_threshold, int32_t* fallback_cells, int32_t* cells_per_frame);

/*
 * Grid-bucketed FAST -- Grider_FAST::perform_griding(img, pts, num_features, grid_x, grid_y, threshold, nonmaxSuppression)
 * (include/Grider_FAST.h:81-137; the OpenVINS helper north_star names; its only call, src/Tracking.cc:940, is commented
 * out in the reference, so this is the alternative bucketing mode).  Host buffers; keypoints come out ROI-major, inside
 * a ROI by response descending, then y, then x (the reference's std::sort leaves ties unspecified).  Uses the
 * extractor handle's device scratch; the image must fit max_width x max_height.
 */
int uvo_grider_fast(uvo_extractor* h, const uint8_t* img, int width, int height, ptrdiff_t stride, int num_features, int grid_x, int grid_y,
                    int threshold, int nonmax_suppression, uvo_keypoint* out_kp, int cap, int* n_out);

/*
 * cv::CLAHE::apply (8-bit), the pre-processing of Tracking::GrabImage when Enhance = 1 (src/Tracking.cc:425-431: clip limit 4,
 * 12 x 12 tiles): per-tile clipped histogram LUTs on the image extended by REFLECT_101 to a multiple of the tile grid, then the
 * bilinear blend of the four neighbouring LUTs per pixel.  In place is allowed (dst == src, same strides).
 * uvo_clahe: host buffers, one frame; the enhanced frame also stays in the handle's HBM, so that the calls that follow in
 * Tracking::GrabImage need no second upload: uvo_extract(img = NULL, same width / height) extracts from it and
 * uvo_klt_build_pyramid_from_extractor() builds the optical-flow pyramid from it; dst = NULL skips the download altogether.
 * uvo_clahe_batch_device: HBM-resident, enqueued on the stream the next
 * uvo_extract_batch_device() call of this handle will use, so that enhance -> extract needs no synchronisation in between.
 */
int uvo_clahe(uvo_extractor* h, const uint8_t* img, int width, int height, ptrdiff_t stride, double clip_limit, int tiles_x, int tiles_y,
              uint8_t* dst, ptrdiff_t dst_stride);
int uvo_clahe_batch_device(uvo_extractor* h, int batch, const uint8_t* d_imgs, int width, int height, ptrdiff_t stride, ptrdiff_t frame_stride,
                           double clip_limit, int tiles_x, int tiles_y, uint8_t* d_dst, ptrdiff_t dst_stride, ptrdiff_t dst_frame_stride);

/* Stage taps for the parity tests (valid after a completed extract call; host destination buffers). */
int uvo_extractor_level_dims(const uvo_extractor* h, int level, int* width, int* height);
/* padded plane (width+32) x (height+32), tight rows; which: 0 = pyramid level, 1 = blurred level */
int uvo_extractor_read_plane(uvo_extractor* h, int frame, int level, int which, uint8_t* dst);
/* FAST candidates of one level: (x, y, score) int32 triples relative to the (13,13) detection border,
 * unordered; returns count via n */
int uvo_extractor_read_candidates(uvo_extractor* h, int frame, int level, int32_t* dst_xys, int cap, int* n);

/*
 * Per-kernel device timing, measured with HIP events on the handle's stream around every launch made while
 * profiling is enabled (uvo_extractor_profile(h, 1); adds two event records per launch).
 * uvo_extractor_kernel_times() waits for the stream, then reports and clears what was recorded:
 * names: '\n'-separated kernel names into `names` (cap bytes); ms[i]: summed duration of that kernel's launches;
 * launches[i]: number of launches.  Behind the kernels' rows, while `cap` allows, follow spread rows for every kernel with two or more
 * launches: "name:min" / "name:p50" / "name:max" = shortest / median / longest single launch (ms), and "name:period_min" / ":period_p50" /
 * ":period_max" = start-to-start time of consecutive launches in enqueue order across the pipeline lanes (for a kernel launched once per
 * batch: the step period as the device saw it), and "name:period2_min" / ":period2_p50" / ":period2_max" = half the start-to-start time of
 * launches two apart (two pipeline lanes take the batches in turn: one lane's period per step, whatever the phase between the lanes);
 * launches[i] of a spread row = the number of samples behind it.
 */
int uvo_extractor_profile(uvo_extractor* h, int enable);
/* restricts the timing to launches of one kernel (name as reported by uvo_extractor_kernel_times; NULL or "" = all kernels) */
int uvo_extractor_profile_only(uvo_extractor* h, const char* kernel_name);
int uvo_extractor_kernel_times(uvo_extractor* h, char* names, int names_cap, float* ms, int32_t* launches, int cap, int* n);

/* ------------------------------------------------------------------------------------------------
 * Matcher -- replaces the arithmetic + search cores of USLAM::ORBmatcher (include/ORBmatcher.h:41-88,
 * src/ORBmatcher.cc) and the all-pairs knn-2 matcher of include/utils.h:81-111.
 * ---------------------------------------------------------------------------------------------- */
typedef struct uvo_matcher uvo_matcher;

typedef struct uvo_matcher_cfg {
  int32_t max_query, max_train; /* descriptors per set */
  int32_t max_batch;            /* descriptor-set pairs per batched call */
  int32_t max_map_points;       /* uvo_search_by_projection */
  int32_t device;
} uvo_matcher_cfg;

int uvo_matcher_create(const uvo_matcher_cfg* cfg, uvo_matcher** out);
void uvo_matcher_destroy(uvo_matcher* m);
int uvo_matcher_synchronize(uvo_matcher* m);

/*
 * All-pairs 256-bit Hamming knn-2: Utils::ratioMatching's knnMatch(desc1, desc2, knn=2, mask)
 * (include/utils.h:100-101); distance = ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1794-1810).
 * mask: nq x nt bytes, non-zero = pair allowed, or NULL.  Ties keep the lower train index.
 * idx = -1, d = 0xFFFF where fewer than one / two train rows are allowed.  Host buffers.
 */
int uvo_hamming_knn2(uvo_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* mask, int32_t* idx0, uint16_t* d0,
                     int32_t* idx1, uint16_t* d1);
/*
 * Batched, HBM-resident: pair p matches d_q[p*q_stride ...] (d_nq[p] rows) against d_t[p*t_stride ...]
 * (d_nt[p] rows); strides in descriptors; outputs [P][max_query].  Enqueues on the matcher's stream.
 * q_stride <= max_query, t_stride <= 65535 (UVO_E_BADARG otherwise).  The counts are device values (an extractor's
 * d_n_out may exceed the slice it could fill): the kernel clamps d_nq[p] to q_stride and d_nt[p] to t_stride.
 */
int uvo_hamming_knn2_batch_device(uvo_matcher* m, int pairs, const uint8_t* d_q, const int32_t* d_nq, int q_stride, const uint8_t* d_t,
                                  const int32_t* d_nt, int t_stride, int32_t* d_idx0, uint16_t* d_d0, int32_t* d_idx1, uint16_t* d_d1);
/* Full nq x nt distance matrix (uint16), the core of MapPoint::ComputeDistinctiveDescriptors
 * (src/MapPoint.cc:236-247).  Host buffers. */
int uvo_hamming_matrix(uvo_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* dist);

/*
 * MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:197-270) for a batch of map points: desc holds the observed
 * descriptors of all points back to back, point p owns rows offsets[p] .. offsets[p+1]-1 (at most 65535 each).  Per point:
 * best_idx = row (relative to offsets[p]) with the least median Hamming distance to the point's rows, itself included,
 * median = sorted[int(0.5*(N-1))], first index on ties; best_median = that median; -1 / -1 for an empty point.  Host buffers.
 */
int uvo_distinctive_descriptors(uvo_matcher* m, const uint8_t* desc, const int32_t* offsets, int npoints, int32_t* best_idx,
                                int32_t* best_median);

/*
 * ORBmatcher::SearchByProjection(FrameKTL&, const vector<MapPoint*>&, th): src/ORBmatcher.cc:49-125, with the
 * frame grid of src/FrameKTL.cc:250-264,359-436 (64 x 48 cells, PosInGrid uses round()).  Host buffers.
 *   frame : kp[n] (undistorted keypoints: x, y, octave are read), desc[n][32], image bounds min/max x/y
 *   map   : per map point the values FrameKTL::isInFrustum left on it (src/FrameKTL.cc:346-352):
 *           proj_x, proj_y, level (mnTrackScaleLevel), view_cos, in_view (mbTrackInView && !isBad), desc[nmp][32]
 *   assigned[n] : in/out, index of the map point held by keypoint i or -1 (reference: F.mvpMapPoints[i] != NULL)
 *   th, nnratio : call-site values (src/Tracking.cc:2222-2228);  scale_factors[nlevels] = F.mvScaleFactors
 * The greedy, order-dependent assignment of the reference is reproduced exactly.
 */
int uvo_search_by_projection(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int min_x, int min_y, int max_x, int max_y,
                             int32_t* assigned, int nmp, const float* proj_x, const float* proj_y, const int32_t* level,
                             const float* view_cos, const uint8_t* in_view, const uint8_t* mp_desc, const float* scale_factors,
                             int nlevels, float th, float nnratio, int* n_matches);

/* ------------------------------------------------------------------------------------------------
 * The other search loops of ORBmatcher.  They all share one shape -- queries visited in a fixed order, a candidate
 * list per query, Hamming distance to every candidate, an acceptance rule, and (except Fuse) targets taken by an
 * earlier query skipped by later ones -- so two generic entry points carry them (candidates from grid windows, or
 * given by the caller), and the reference-named entry points below are thin marshalling layers over those.
 * The order-dependent loops are reproduced exactly (see csrc/match_engine.hip).  Host buffers throughout.
 * ---------------------------------------------------------------------------------------------- */
enum {
  UVO_RULE_BEST_RATIO_SAME_LEVEL = 0, /* SearchByProjection(F, vpMapPoints, th): src/ORBmatcher.cc:96-123 */
  UVO_RULE_BEST_ONLY = 1,             /* SearchByProjection(F, pKF, ...) :1681-1701; Fuse :1084-1119: d <= max_dist */
  UVO_RULE_BEST_RATIO_LE = 2,         /* SearchByBoW(pKF, F, ...) :194-218: d <= max_dist && d < nnratio * second */
  UVO_RULE_BEST_RATIO_LT = 3,         /* SearchByBoW(pKF1, pKF2, ...) :762-788: d < max_dist && d < nnratio * second */
  UVO_RULE_TRIANGULATION = 4,         /* SearchForTriangulation :893-935: d <= max_dist, sorted, d <= 2*best, epipolar */
  /* the members of ORBmatcher that have no caller in the reference */
  UVO_RULE_BEST_RATIO_LEQ = 5,        /* WindowSearch :409-516, SearchByProjection(F1, F2, windowSize) :519-594: d <= nnratio * second
                                         (second = INT_MAX when there is none) && d <= max_dist; both use exclusive = 1 */
  UVO_RULE_INIT_STEAL = 6             /* SearchForInitialization :598-713: candidates whose target is currently matched at <= d are
                                         skipped, d <= max_dist && d < nnratio * second, and a later query with a strictly smaller
                                         distance takes a target over (`exclusive` is ignored); match[i] >= 0 only for queries that
                                         still hold their target at the end; the rotation histogram counts every accept (:662-670) */
};
typedef struct uvo_match_rule {
  int32_t rule;              /* UVO_RULE_* */
  int32_t max_dist;          /* TH_HIGH (100), TH_LOW (50) or the caller's ORBdist */
  float nn_ratio;            /* mfNNratio */
  int32_t exclusive;         /* 1: a target accepted by an earlier query is skipped by later ones */
  int32_t check_orientation; /* mbCheckOrientation: rotation histogram + ComputeThreeMaxima (:1748-1789) */
} uvo_match_rule;
/* ORBmatcher::CheckDistEpipolarLine (:136-153): l = x1' F12; dsqr = (l . x2)^2 / (a^2+b^2) < 3.84 * sigma2[octave2] */
typedef struct uvo_epipolar {
  float f12[9];         /* row-major 3x3 */
  const float* q_x;     /* query keypoint coordinates (kp1.pt) */
  const float* q_y;
  const float* t_x;     /* target keypoint coordinates (kp2.pt) */
  const float* t_y;
  const float* sigma2;  /* pKF2->GetSigma2(level), nlevels entries */
  int32_t nlevels;
} uvo_epipolar;

/*
 * Candidates from grid windows: FrameKTL::GetFeaturesInArea (src/FrameKTL.cc:359-424) / KeyFrame::GetFeaturesInArea
 * (src/KeyFrame.cc:952-992) over the 64 x 48 grid of the target keypoints kp[n].
 *   blocked[n]  : may be NULL; non-zero = target unavailable from the start (e.g. F.mvpMapPoints[i] != NULL)
 *   query i     : window centre (qx, qy), radius qr, level filter [qmin_level, qmax_level] with the reference's meaning
 *                 (-1,-1 = no filter; equal = that level only), qvalid (0 = query skipped), qdesc, qangle (read only
 *                 when rule->check_orientation; the target angle is kp[].angle)
 *   match[nq]   : target index or -1;  dist[nq] : its distance or -1;  *n_matches : number of matches
 */
int uvo_match_windows(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, const uint8_t* blocked, int min_x, int min_y,
                      int max_x, int max_y, int nq, const float* qx, const float* qy, const float* qr, const int32_t* qmin_level,
                      const int32_t* qmax_level, const uint8_t* qvalid, const uint8_t* qdesc, const float* qangle, const uvo_match_rule* rule,
                      int32_t* match, int32_t* dist, int* n_matches);
/*
 * Candidates given by the caller: query i owns cand_idx[cand_start[i] .. cand_start[i+1]) (target indices, in the order
 * the reference would visit them).  tlevel / tangle / qangle / tblocked / epi may be NULL when the rule does not read them.
 */
int uvo_match_groups(uvo_matcher* m, int nq, const uint8_t* qdesc, const float* qangle, int nt, const uint8_t* tdesc, const float* tangle,
                     const int32_t* tlevel, const uint8_t* tblocked, const int32_t* cand_start, const int32_t* cand_idx,
                     const uvo_epipolar* epi, const uvo_match_rule* rule, int32_t* match, int32_t* dist, int* n_matches);

/*
 * ORBmatcher::SearchByProjection(FrameKTL& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist): :1622-1746.
 * Per key-frame map point i the caller passes what :1647-1672 computes (uvo_project_points() does it on the device):
 * valid[i] (non-null, !isBad, not already found, projection inside the image), u, v, predicted level; kf_angle[i] =
 * pKF->GetKeyPointUn(i).angle.  Window radius th * scale_factors[level], levels [level-1, level+1], best distance
 * <= orb_dist, first-come exclusivity over the frame keypoints, rotation histogram when check_orientation.
 *   assigned[n] : in/out, index of the map point held by frame keypoint k or -1 (F.mvpMapPoints[k])
 */
int uvo_search_by_projection_kf(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int min_x, int min_y, int max_x, int max_y,
                                int32_t* assigned, int nmp, const float* u, const float* v, const int32_t* level, const uint8_t* valid,
                                const uint8_t* mp_desc, const float* kf_angle, const float* scale_factors, int nlevels, float th, int orb_dist,
                                int check_orientation, int* n_matches);

/*
 * A DBoW2::FeatureVector (std::map<NodeId, vector<unsigned>>) in flat form: node ids ascending, features of node j =
 * feat[start[j] .. start[j+1]).
 */
typedef struct uvo_feature_vector {
  const uint32_t* node;
  const int32_t* start;
  const int32_t* feat;
  int32_t n_nodes;
} uvo_feature_vector;

/*
 * ORBmatcher::SearchByBoW(KeyFrame* pKF, FrameKTL& F, vpMapPointMatches) :155-284 (kf_kf = 0) and
 * SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) :715-850 (kf_kf = 1).
 *   side 1 (queries): n1 keypoints, desc1, angle1, usable1[n1] = pMP1 != NULL && !isBad
 *   side 2 (targets): n2 keypoints, desc2, angle2, usable2[n2] = kf_kf ? (pMP2 != NULL && !isBad) : 1 (NULL = all usable)
 *   match12[n1] : target index or -1 (the reference stores the map point: vpMapPoints...[that index])
 */
int uvo_search_by_bow(uvo_matcher* m, int kf_kf, const uvo_feature_vector* fv1, int n1, const uint8_t* desc1, const float* angle1,
                      const uint8_t* usable1, const uvo_feature_vector* fv2, int n2, const uint8_t* desc2, const float* angle2,
                      const uint8_t* usable2, float nnratio, int check_orientation, int32_t* match12, int* n_matches);

/*
 * ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, ...) :852-1014.  kp1 / kp2 = undistorted keypoints (x, y, octave,
 * angle are read), has_mp1 / has_mp2 = "keypoint already has a map point", f12 row-major, sigma2 = pKF2 level sigmas.
 *   match12[n1] : index into kp2 or -1 (vMatchedPairs = the pairs with match12[i] >= 0, ascending i)
 */
int uvo_search_for_triangulation(uvo_matcher* m, const uvo_feature_vector* fv1, const uvo_keypoint* kp1, int n1, const uint8_t* desc1,
                                 const uint8_t* has_mp1, const uvo_feature_vector* fv2, const uvo_keypoint* kp2, int n2, const uint8_t* desc2,
                                 const uint8_t* has_mp2, const float* f12, const float* sigma2, int nlevels, int check_orientation,
                                 int32_t* match12, int* n_matches);

/*
 * Search core of ORBmatcher::Fuse(pKF, vpMapPoints, th) :1016-1134 and Fuse(pKF, Scw, ...) :1136-1265: per candidate map
 * point i (valid[i] = passed the projection tests :1037-1076, see uvo_project_points) the best key-frame keypoint in the
 * window of radius th * scale_factors[level] on levels [level-1, level] with distance <= TH_LOW; no exclusivity.
 *   best_idx[nmp] / best_dist[nmp] : keypoint index and distance, or -1.  The map mutation (:1104-1118) stays with the caller.
 */
int uvo_fuse(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int min_x, int min_y, int max_x, int max_y, int nmp,
             const float* u, const float* v, const int32_t* level, const uint8_t* valid, const uint8_t* mp_desc, const float* scale_factors,
             int nlevels, float th, int32_t* best_idx, int32_t* best_dist);

/*
 * Projection prologues of the search loops: what the reference computes per map point before it queries the grid.
 *   UVO_PROJECT_FRUSTUM  : FrameKTL::isInFrustum (src/FrameKTL.cc:299-357) + MapPoint::PredictScale (src/MapPoint.cc:373-388)
 *                          -> valid (mbTrackInView), u, v (mTrackProjX/Y), level (mnTrackScaleLevel), view_cos (mTrackViewCos):
 *                          the inputs of uvo_search_by_projection
 *   UVO_PROJECT_KF_RELOC : SearchByProjection(CurrentFrame, pKF, ...) src/ORBmatcher.cc:1647-1670 -> the inputs of
 *                          uvo_search_by_projection_kf (camera centre derived as -Rcw^T tcw like :1628; `ow` is ignored)
 *   UVO_PROJECT_FUSE     : Fuse src/ORBmatcher.cc:1037-1075 -> the inputs of uvo_fuse
 * Arithmetic follows the reference expression by expression (fp32 products, the double intermediates of `1.0/z`,
 * cv::norm and cv::Mat::dot, libm logf); OpenCV's 3x3 * 3x1 product is taken as its small-matrix path (fp32 row sums, the
 * `+ t` in double) -- an unpinned assumption, stated in DESIGN.md.  log(scaleFactor) is the intended constant
 * (the reference reads it before it is initialised).
 *   usable[i]      : 0 = skip (NULL / isBad / already found / already in the key frame ...), may be NULL
 *   min_distance_inv / max_distance_inv : GetMinDistanceInvariance() / GetMaxDistanceInvariance() of every point
 *                    (max_distance_inv is not read by KF_RELOC; the PIXEL modes read neither and accept NULL); max_distance: the raw mfMaxDistance member that
 *                    MapPoint::PredictScale divides by (FRUSTUM only); normal: GetNormal(), modes FRUSTUM and FUSE
 */
enum {
  UVO_PROJECT_FRUSTUM = 0,
  UVO_PROJECT_KF_RELOC = 1,
  UVO_PROJECT_FUSE = 2,
  /* the two caller-less projection searches: u, v only; the caller supplies the level (the keypoint's octave) */
  UVO_PROJECT_PIXEL_BOUNDED = 3, /* SearchByProjection(CurrentFrame, LastFrame, th) :1530-1545: valid = inside [min_x, max_x] x [min_y, max_y] */
  UVO_PROJECT_PIXEL = 4          /* SearchByProjection(F1, F2, windowSize) :541-550: no test (not even the depth's sign) */
};
typedef struct uvo_camera_pose {
  float rcw[9]; /* row-major world -> camera rotation */
  float tcw[3];
  float ow[3];  /* camera centre in the world (mOw / GetCameraCenter()) */
  float fx, fy, cx, cy;
  float min_x, max_x, min_y, max_y; /* mnMinX, mnMaxX, mnMinY, mnMaxY */
} uvo_camera_pose;
int uvo_project_points(uvo_matcher* m, int mode, const uvo_camera_pose* cam, int npts, const float* xyz, const float* normal,
                       const float* min_distance_inv, const float* max_distance_inv, const float* max_distance, const uint8_t* usable,
                       const float* scale_factors, int nlevels, float scale_factor, float viewing_cos_limit, uint8_t* valid, float* u, float* v,
                       int32_t* level, float* view_cos);

/*
 * Batched forms of the two matcher loops of the LocalMapping thread: one device round trip for the whole loop, results identical to
 * the single calls made one after the other.
 *
 * LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:1058-1180) calls SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, ...) for
 * up to 20 neighbour key frames; between two calls it triangulates the pair's matches and the accepted ones get map points
 * (mpCurrentKeyFrame->AddMapPoint, :1177), which removes those features from the later pairs (`if(pMP1) continue;`,
 * src/ORBmatcher.cc:885-889).  uvo_search_for_triangulation_batch() computes the descriptor distances and the epipolar test of EVERY
 * pair in one launch (one upload, one host wait) for the features of key frame 1 that have no map point yet, and keeps the candidate
 * lists in the handle; uvo_search_for_triangulation_next(pair, has_mp1 as it is NOW, ...) replays the reference's acceptance loop
 * (:886-984: candidates still free, distance <= TH_LOW, sorted, walk to round(2 * best), first one on the epipolar line; rotation
 * histogram) for that pair on the host -- no device work.  Pairs may be replayed in any order and more than once; the caller passes
 * the has_mp1 the reference would see at that point (a feature may only GAIN a map point while a batch is alive: UVO_E_BADARG if one
 * lost it).  A new batch replaces the old one.
 */
typedef struct uvo_triangulation_pair {
  const uvo_feature_vector* fv2; /* pKF2->GetFeatureVector() */
  const uvo_keypoint* kp2;       /* pKF2->GetKeyPointsUn() */
  int32_t n2;
  const uint8_t* desc2;          /* [n2][32] */
  const uint8_t* has_mp2;        /* [n2] pKF2->GetMapPointMatches()[k] != NULL */
  float f12[9];                  /* row-major */
  const float* sigma2;           /* [nlevels] pKF2->GetSigma2(level) */
  int32_t nlevels;
} uvo_triangulation_pair;
int uvo_search_for_triangulation_batch(uvo_matcher* m, const uvo_feature_vector* fv1, const uvo_keypoint* kp1, int n1, const uint8_t* desc1,
                                       const uint8_t* has_mp1, int n_pairs, const uvo_triangulation_pair* pairs);
int uvo_search_for_triangulation_next(uvo_matcher* m, int pair, const uint8_t* has_mp1_now, int check_orientation, int32_t* match12,
                                      int* n_matches);
/*
 * LocalMapping::SearchInNeighbors (src/LocalMapping.cc:1228-1236) calls Fuse(pKFi, vpMapPointMatches) for every target key frame.  The
 * search core of Fuse has no exclusivity among the map points, so the projection tests (:1037-1075, with each target's own pose) and the
 * best key point (:1077-1101) of EVERY (target, map point) are computed in one pass: the map points are uploaded once, each target adds
 * its grid + projection + window walk to the stream, one download, one host wait.
 *   usable[nmp]            : 0 = the point is NULL (never searched); NULL = all usable.  The tests that depend on the map as the loop
 *                            mutates it -- isBad(), IsInKeyFrame(pKFi), :1031-1035 -- stay with the caller, who discards the results of
 *                            points that fail them when target i's turn comes (the search of a point depends on nothing else).
 *   best_idx / best_dist   : [n_targets][nmp] key point of target t and distance, or -1 (as uvo_fuse).
 * One thing the caller must watch: pMP->Replace(pMPinKF) (:1107) recomputes pMPinKF's descriptor; if pMPinKF is itself one of the nmp
 * points, its rows of the LATER targets were computed with the old descriptor and must be redone (uvo_project_points + uvo_fuse for
 * that point).  include/uvo/compat/ORBmatcher.h FuseTargets() does exactly that.
 */
typedef struct uvo_fuse_target {
  const uvo_keypoint* kp; /* pKF->GetKeyPointUn(k) for all k */
  int32_t n;
  const uint8_t* desc;    /* [n][32] */
  int32_t min_x, min_y, max_x, max_y; /* mnMinX .. mnMaxY */
  uvo_camera_pose cam;    /* pose, intrinsics and image bounds of the target */
  const float* scale_factors;
  int32_t nlevels;
} uvo_fuse_target;
int uvo_fuse_batch(uvo_matcher* m, int n_targets, const uvo_fuse_target* targets, int nmp, const float* xyz, const float* normal,
                   const float* min_distance_inv, const float* max_distance_inv, const uint8_t* usable, const uint8_t* mp_desc, float th,
                   int32_t* best_idx, int32_t* best_dist);


/*
 * Tracking::SearchReferencePointsInFrustum (src/Tracking.cc:2176-2230) as one call: FrameKTL::isInFrustum(pMP, viewing_cos_limit) on
 * every local map point (UVO_PROJECT_FRUSTUM above), then SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th) (:49-125) on the
 * points in view -- same results as uvo_project_points followed by uvo_search_by_projection, but the inputs travel as one block,
 * the projections stay on the device and the host waits once.  Frame side: kp / desc / assigned as in uvo_search_by_projection; the
 * grid bounds are cam->min_x .. max_y.  Map side: as in uvo_project_points, plus the representative descriptors mp_desc[npts][32].
 *   usable[i]  : 0 for points the caller's loop skips (:2207-2210: seen in this frame already, or bad); NULL = all usable
 *   in_view, proj_x, proj_y, level, view_cos : optional outputs = mbTrackInView, mTrackProjX/Y, mnTrackScaleLevel, mTrackViewCos
 *   *n_to_match = number of points in view (nToMatch; the reference skips the search when it is 0 -- so does this call, in effect)
 */
int uvo_search_points_in_frustum(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int32_t* assigned, const uvo_camera_pose* cam,
                                 int npts, const float* xyz, const float* normal, const float* min_distance_inv, const float* max_distance_inv,
                                 const float* max_distance, const uint8_t* usable, const uint8_t* mp_desc, const float* scale_factors, int nlevels,
                                 float scale_factor, float viewing_cos_limit, float th, float nnratio, uint8_t* in_view, float* proj_x,
                                 float* proj_y, int32_t* level, float* view_cos, int* n_to_match, int* n_matches);

/*
 * Loop-closing (Sim3) forms of the search loops.
 *
 * uvo_sim3_decompose: the head of SearchByProjection(pKF, Scw, ...) src/ORBmatcher.cc:299-303 and Fuse(pKF, Scw, ...) :1145-1149:
 *   scw = |row 0 of sR|, Rcw = sR / scw, tcw = s t / scw, Ow = -Rcw^T tcw, written to cam->rcw / tcw / ow (the other members are
 *   left alone).  scw_mat: 3 rows of row_stride floats (a 4x4 or 3x4 row-major Scw).  Host arithmetic, evaluated the way OpenCV
 *   evaluates those cv::Mat expressions (double dot product, multiplication by the float reciprocal, double-accumulating gemm).
 *   With the result, uvo_project_points(UVO_PROJECT_FUSE) is the per-point prologue of both members (:315-361 / :1161-1207 are
 *   the same tests as :1037-1075).
 * uvo_sim3_relative: SearchBySim3 :1284-1287: s_r12 = s12 R12, s_r21 = (1/s12) R12^T, t21 = -s_r21 t12 (3x3 row-major, 3-vectors).
 * uvo_project_sim3: the per-point prologue of either direction of SearchBySim3 (:1323-1359 / :1403-1441): world -> the owning
 *   key frame's camera (r_own, t_own) -> the other camera (s_r, t) -> pixel (cam_other: fx, fy, cx, cy and the image bounds are
 *   read), positive depth, KeyFrame::IsInImage, distance inside [min_distance_inv, max_distance_inv], level by lower_bound over
 *   scale_factors.  usable[i] = point exists, not already matched, not bad (may be NULL).
 * uvo_search_by_projection_sim3: :357-398 on the projected candidates: best unmatched key point within th * scale_factors[level]
 *   on levels [level-1, level], accepted at <= TH_LOW, first come first served in candidate order.
 *   matched[n]: in: >= 0 where vpMatched[idx] is set; out: newly matched key points hold the candidate's index.
 * uvo_search_by_sim3: :1361-1504: both directions (best key point of the other frame on levels [level-1, level], <= TH_HIGH,
 *   no exclusivity) and the agreement check.  *12 arrays have n1 entries (KF1's map points, projected into KF2, with their
 *   representative descriptors mp_desc1), *21 arrays n2 entries.  bounds = {mnMinX, mnMinY, mnMaxX, mnMaxY} of each key frame.
 *   match12[n1]: index into KF2 or -1.
 */
int uvo_sim3_decompose(const float* scw_mat, int row_stride, uvo_camera_pose* cam);
int uvo_sim3_relative(float s12, const float* r12, const float* t12, float* s_r12, float* s_r21, float* t21);
int uvo_project_sim3(uvo_matcher* m, const float* r_own, const float* t_own, const float* s_r, const float* t, const uvo_camera_pose* cam_other,
                     int npts, const float* xyz, const float* min_distance_inv, const float* max_distance_inv, const uint8_t* usable,
                     const float* scale_factors, int nlevels, uint8_t* valid, float* u, float* v, int32_t* level);
int uvo_search_by_projection_sim3(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int min_x, int min_y, int max_x, int max_y,
                                  int32_t* matched, int nmp, const float* u, const float* v, const int32_t* level, const uint8_t* valid,
                                  const uint8_t* mp_desc, const float* scale_factors, int nlevels, int th, int* n_matches);
int uvo_search_by_sim3(uvo_matcher* m, const uvo_keypoint* kp1, int n1, const uint8_t* desc1, const int32_t* bounds1, const uvo_keypoint* kp2,
                       int n2, const uint8_t* desc2, const int32_t* bounds2, const float* u12, const float* v12, const int32_t* level12,
                       const uint8_t* valid12, const uint8_t* mp_desc1, const float* u21, const float* v21, const int32_t* level21,
                       const uint8_t* valid21, const uint8_t* mp_desc2, const float* scale_factors1, int nlevels1, const float* scale_factors2,
                       int nlevels2, float th, int32_t* match12, int* n_found);

/* ------------------------------------------------------------------------------------------------
 * Bag-of-words transform: DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, BowVector&, FeatureVector&,
 * levelsup) (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1125-1188, per-feature descent :1207-1258), as called by
 * FrameKTL::ComputeBoW (src/FrameKTL.cc:439-446) and KeyFrame::ComputeBoW (src/KeyFrame.cc:203-210) with levelsup = 4.
 * The tree descent (Hamming distance to every child, first minimum wins) runs on the device; the two std::map containers are
 * assembled on the host in the reference's insertion order.  The feature vector comes out in the flat form
 * uvo_search_by_bow / uvo_search_for_triangulation take.
 * ---------------------------------------------------------------------------------------------- */
typedef struct uvo_vocabulary uvo_vocabulary;
typedef struct uvo_vocabulary_desc {
  int32_t n_nodes;            /* m_nodes.size(); node 0 is the root */
  const int32_t* child_start; /* [n_nodes + 1]: children of node i = children[child_start[i] .. child_start[i+1]), in m_nodes[i].children order */
  const int32_t* children;
  const uint8_t* descriptor;  /* [n_nodes][32]: m_nodes[i].descriptor (row 0 unused) */
  const int32_t* word_id;     /* [n_nodes]: m_nodes[i].word_id (read for leaves) */
  const double* weight;       /* [n_nodes]: m_nodes[i].weight */
  int32_t L;                  /* m_L */
  int32_t weighting;          /* DBoW2::WeightingType: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY */
  int32_t normalize;          /* GeneralScoring::mustNormalize: 0 = no (DotProduct), 1 = L1, 2 = L2 */
  int32_t device;
} uvo_vocabulary_desc;
int uvo_vocabulary_create(const uvo_vocabulary_desc* desc, uvo_vocabulary** out);
void uvo_vocabulary_destroy(uvo_vocabulary* voc);
/*
 * desc[n][32] -> per feature: word_id, word_weight, node_id (node reached at level L - levelsup; 0 when that level is <= 0;
 * the leaf itself when the descent ends above that level -- the reference leaves it uninitialised there);
 * BowVector as (bow_id ascending, bow_value), *n_bow entries; FeatureVector as fv_node ascending, fv_start[*n_fv + 1],
 * fv_feat (feature indices, ascending inside a node).  Per-feature arrays may be NULL.  Host buffers.
 */
int uvo_bow_transform(uvo_vocabulary* voc, const uint8_t* desc, int n, int levelsup, int32_t* word_id, double* word_weight, int32_t* node_id,
                      uint32_t* bow_id, double* bow_value, int bow_cap, int* n_bow, uint32_t* fv_node, int32_t* fv_start, int32_t* fv_feat,
                      int fv_cap, int* n_fv);

/*
 * haloc::Hash::getHash (src/hash.cpp:57-85; KeyFrame::ComputeHaloc src/KeyFrame.cc:318-329): for every projection vector r_i and
 * descriptor column c, hash[i*32 + c] = (sum over rows m of r_i[m] * (float)desc[m][c]) / (float)n, accumulated in fp32 in row
 * order (one thread per output walks the rows, so the rounding sequence is the reference's).  proj: [num_proj][proj_stride]
 * floats (the reference builds them from time(NULL), so they are an input here), proj_stride >= n.  Host buffers.
 */
int uvo_haloc_hash(uvo_matcher* m, const float* proj, int num_proj, int proj_stride, const uint8_t* desc, int n, float* hash);

/*
 * Device-side ordering between the two handles' streams (no host synchronisation): work enqueued on the
 * matcher after uvo_matcher_wait_extractor() starts only when everything enqueued on the extractor so far
 * has finished, and vice versa.  Used when descriptors produced by uvo_extract_batch_device() feed
 * uvo_hamming_knn2_batch_device() directly in HBM.
 */
int uvo_matcher_wait_extractor(uvo_matcher* m, uvo_extractor* h);
int uvo_extractor_wait_matcher(uvo_extractor* h, uvo_matcher* m);
/*
 * The cheaper form of the same ordering: from this call on the matcher enqueues its work in the stream of the extractor's current
 * pipeline lane (the lane of the most recent batch call), directly behind that batch's kernels and in front of whatever the lane
 * runs next -- no events, no hand-off between queues (each costs tens of microseconds; a lane of the batch-256 pipeline idled
 * 0.3 ms per batch on the two hand-offs around uvo_hamming_knn2_batch_device).  One call is enough: the extractor keeps a list of the
 * matchers attached to it and moves them along whenever a batch goes to another pipeline lane, so the matcher always works behind the
 * most recent batch (calling it again after a batch is harmless).  h = NULL: back to the matcher's own stream.  While attached, calls
 * that wait for the matcher's stream (every host-buffer entry point) wait for that lane.  Either handle may be destroyed first: a
 * destroyed extractor hands its matchers back to their own streams, a destroyed matcher leaves the list.
 * ORDERING CONTRACT: an attached matcher is ordered behind the MOST RECENT batch only.  With pipeline depth >= 2, work that reads the
 * results of batch N must be enqueued before batch N + 1 is (extract N, match N, extract N + 1, match N + 1 ...); a caller that
 * enqueues two batches and then matches the first must order that match itself (uvo_matcher_wait_extractor before batch N + 1, or
 * uvo_extractor_synchronize).  The handles are not thread-safe against each other: attach / detach / destroy and the batch calls of
 * one extractor + its matchers belong to one host thread at a time (the follower list itself is locked).
 */
int uvo_matcher_attach_extractor(uvo_matcher* m, uvo_extractor* h);
/* per-kernel timing of the matcher, same contract as uvo_extractor_profile / uvo_extractor_kernel_times */
int uvo_matcher_profile(uvo_matcher* m, int enable);
int uvo_matcher_kernel_times(uvo_matcher* m, char* names, int names_cap, float* ms, int32_t* launches, int cap, int* n);

/* ------------------------------------------------------------------------------------------------
 * KLT step in front of the extractor: cv::buildOpticalFlowPyramid (src/FrameKTL.cc:76) and cv::calcOpticalFlowPyrLK with
 * OPTFLOW_USE_INITIAL_FLOW + OPTFLOW_LK_GET_MIN_EIGENVALS (src/Tracking.cc:1046-1047).  Pyramids (image levels with a
 * winSize REFLECT_101 border + Scharr derivatives) stay on the device in numbered slots, so a frame's pyramid is built once
 * and tracked against twice (as previous, then as next).  Integer stages are exact; the tracker's float sums are reduced
 * across a wavefront, i.e. associated differently from a raster-order CPU loop (positions agree to float rounding).
 * ---------------------------------------------------------------------------------------------- */
typedef struct uvo_klt uvo_klt;
typedef struct uvo_klt_cfg {
  int32_t max_width, max_height;
  int32_t max_level;              /* mPyr_Levels: levels 0..max_level (fewer when a level would not exceed the window) */
  int32_t win_width, win_height;  /* mWin_Size, e.g. 21 x 21; at most 1024 pixels */
  int32_t max_points;
  int32_t slots;                  /* pyramids kept on the device (>= 2) */
  int32_t device;
} uvo_klt_cfg;
int uvo_klt_create(const uvo_klt_cfg* cfg, uvo_klt** out);
void uvo_klt_destroy(uvo_klt* k);
int uvo_klt_build_pyramid(uvo_klt* k, int slot, const uint8_t* img, int width, int height, ptrdiff_t stride, int* levels_built);
/* HBM-resident chain: the image is the result of the extractor handle's last uvo_clahe() call (see there), no upload */
int uvo_klt_build_pyramid_from_extractor(uvo_klt* k, int slot, uvo_extractor* h, int* levels_built);
/* test tap: level without its border; img [h][w] u8, deriv [h][w][2] int16 (either may be NULL) */
int uvo_klt_read_level(uvo_klt* k, int slot, int level, uint8_t* img, int16_t* deriv, int* width, int* height);
/* prev_pts / next_pts: n x (x, y) float32; next_pts in = initial flow, out = tracked positions; status[n], err[n] (min eigenvalue).
 * max_count / epsilon: the TermCriteria (30, 0.01 at the call site); min_eig_threshold: 1e-4 (OpenCV default). */
int uvo_klt_track(uvo_klt* k, int prev_slot, int next_slot, const float* prev_pts, float* next_pts, int n, int max_level, int max_count,
                  double epsilon, double min_eig_threshold, uint8_t* status, float* err);
/*
 * Tracking::undistort_point (src/Tracking.cc:1265-1283), the step between the tracker and RANSAC (:1049-1053): for the pin-hole model
 * cv::undistortPoints(pt, pt, mK, mDistCoef, cv::Mat(), mK), for Fisheye_Cam cv::fisheye::undistortPoints(pt, pt, mK, mDistCoef,
 * cv::Mat(), mK).  dist: k1 k2 p1 p2 [k3 [k4 k5 k6]] (n_dist 0..8) resp. the four fisheye coefficients (n_dist <= 4).  Double
 * arithmetic like OpenCV's; results are float pixel coordinates.
 *   uvo_undistort_points      : n points, host in / host out (n <= 2 * max_points of the handle)
 *   uvo_klt_track_undistorted : uvo_klt_track + the undistortion of both point sets (prev_un, next_un: n x 2 floats) on the device --
 *                               the loop of :1049-1053 as part of the same call, one upload and one download
 */
typedef struct uvo_camera_model {
  float fx, fy, cx, cy; /* mK (CV_32F) */
  float dist[8];        /* mDistCoef */
  int32_t n_dist;
  int32_t fisheye;      /* Fisheye_Cam */
} uvo_camera_model;
int uvo_undistort_points(uvo_klt* k, const uvo_camera_model* cam, const float* pts, int n, float* out);
int uvo_klt_track_undistorted(uvo_klt* k, int prev_slot, int next_slot, const float* prev_pts, float* next_pts, int n, int max_level, int max_count,
                              double epsilon, double min_eig_threshold, const uvo_camera_model* cam, uint8_t* status, float* err, float* prev_un,
                              float* next_un);

/* last HIP / argument error text for the calling thread's most recent failing call (never NULL) */
const char* uvo_last_error(void);
/* library + device description, e.g. "uvo 0.1 gfx950 AMD Instinct MI355X" */
int uvo_device_info(int device, char* dst, int cap);

#ifdef __cplusplus
}
#endif
#endif /* UVO_UVO_H_ */
