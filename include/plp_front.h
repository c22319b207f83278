/* plp_front.h — C ABI of libplp_front.so, the MI355X-native feature front-end and matcher
 * that replaces the per-frame hot path of Structure-PLP-SLAM (src/PLPSLAM/feature,
 * src/PLPSLAM/match).  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repository root).  All functions return a plp_status and never throw.
 *
 * Pointer naming: `h_` / unprefixed = host memory, `d_` = device (HBM) memory of the
 * context's GPU.  Batched `_device` entry points are asynchronous on `hip_stream`
 * (a hipStream_t passed as void*; NULL = HIP's default stream, so that callers working on the
 * default stream -- e.g. PyTorch's current stream -- stay ordered) unless stated otherwise.  The host-pointer entry
 * points use the context's own non-blocking stream and synchronise it before returning.
 */
#ifndef PLP_FRONT_H
#define PLP_FRONT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum plp_status {
    PLP_OK = 0,
    PLP_ERR_INVALID_ARG = 1,   /* what the reference reports via std::runtime_error / assert */
    PLP_ERR_NO_DEVICE = 2,     /* no usable MI355X / HIP runtime failure at create time */
    PLP_ERR_HIP = 3,           /* a HIP call failed; see plp_last_error() */
    PLP_ERR_CAPACITY = 4,      /* caller buffer too small (n_out still holds the needed size) */
    PLP_ERR_OVERFLOW = 5,      /* an internal per-level candidate buffer overflowed */
    PLP_ERR_UNSUPPORTED = 6
} plp_status;

const char* plp_strerror(plp_status s);
const char* plp_last_error(void);      /* thread-local detail string of the last failure */
int plp_version(void);                 /* 100*major + minor */
int plp_device_count(void);            /* number of visible HIP devices (0 on a CPU-only box) */

/* ------------------------------------------------------------------------------------------
 * Key point record: field-for-field cv::KeyPoint (28 bytes), the element type of
 * `std::vector<cv::KeyPoint>& keypts` in feature::orb_extractor::extract
 * (src/PLPSLAM/feature/orb_extractor.h:53-54).
 * ---------------------------------------------------------------------------------------- */
typedef struct plp_keypoint {
    float x, y;        /* pt, in level-0 pixel coordinates                                    */
    float size;        /* (unsigned)(31 * scale_factor[octave])     orb_extractor.cc:448,457  */
    float angle;       /* degrees in [0,360), intensity-centroid    orb_extractor.cc:708-735  */
    float response;    /* FAST-9/16 corner score                                              */
    int32_t octave;    /* pyramid level                                                        */
    int32_t class_id;  /* always -1                                                            */
} plp_keypoint;

/* ------------------------------------------------------------------------------------------
 * ORB extractor  — replaces feature::orb_extractor (src/PLPSLAM/feature/orb_extractor.h:38-176)
 * and feature::orb_params (src/PLPSLAM/feature/orb_params.h:34-73).
 * ---------------------------------------------------------------------------------------- */
typedef struct plp_orb plp_orb;   /* one per feature::orb_extractor instance; owns a HIP stream */

typedef struct plp_orb_params {   /* orb_params.h:52-60, defaults 2000 / 1.2 / 8 / 20 / 7 */
    uint32_t max_num_keypts;
    float scale_factor;
    uint32_t num_levels;
    uint32_t ini_fast_thr;
    uint32_t min_fast_thr;
    const float* mask_rects;      /* n_mask_rects x [x_min/cols, x_max/cols, y_min/rows, y_max/rows] */
    int32_t n_mask_rects;
} plp_orb_params;

void plp_orb_default_params(plp_orb_params* p);

/* orb_extractor::orb_extractor(const orb_params&)  (orb_extractor.cc:66-71).  Validation failures
 * that make orb_params throw std::runtime_error (orb_params.cc:40-54) return PLP_ERR_INVALID_ARG. */
plp_status plp_orb_create(const plp_orb_params* params, int device, plp_orb** out);
void plp_orb_destroy(plp_orb* ctx);

typedef enum plp_orb_param_id {
    PLP_ORB_MAX_NUM_KEYPOINTS = 0,     /* get/set_max_num_keypoints        orb_extractor.cc:162-171 */
    PLP_ORB_SCALE_FACTOR = 1,          /* get/set_scale_factor             :173-182 */
    PLP_ORB_NUM_SCALE_LEVELS = 2,      /* get/set_num_scale_levels         :184-193 */
    PLP_ORB_INITIAL_FAST_THRESHOLD = 3,/* get/set_initial_fast_threshold   :195-203 */
    PLP_ORB_MINIMUM_FAST_THRESHOLD = 4 /* get/set_minimum_fast_threshold   :205-213 */
} plp_orb_param_id;
plp_status plp_orb_set_param(plp_orb* ctx, plp_orb_param_id id, double value);  /* re-runs initialize() like the setters */
plp_status plp_orb_get_param(const plp_orb* ctx, plp_orb_param_id id, double* value);

/* get_scale_factors / get_inv_scale_factors / get_level_sigma_sq / get_inv_level_sigma_sq
 * (orb_extractor.cc:215-233); each output array holds num_levels floats (NULL = skip).
 * quota = num_keypts_per_level_ (orb_extractor.cc:245-253). */
plp_status plp_orb_get_tables(const plp_orb* ctx, int32_t* n_levels, float* scale_factors,
                              float* inv_scale_factors, float* level_sigma_sq, float* inv_level_sigma_sq,
                              uint32_t* quota);

/* orb_extractor::extract(in_image, in_image_mask, keypts, out_descriptors)  (orb_extractor.cc:73-160).
 * Host pointers, synchronous.  img: rows x cols CV_8UC1, `step` bytes per row.  mask: NULL or same
 * size (0 = masked).  kps/desc: caller buffers of `cap` records / cap x 32 bytes; *n_out receives the
 * number of key points (concatenated by level).  An empty image (rows==0||cols==0) is a no-op that
 * leaves *n_out untouched, as the reference does (:76-79). */
plp_status plp_orb_extract(plp_orb* ctx, const uint8_t* img, int32_t rows, int32_t cols, size_t step,
                           const uint8_t* mask, size_t mask_step,
                           plp_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out);

/* Batched replay form of extract(): B frames already resident in HBM, results stay in HBM.
 * d_imgs: B frames, frame b at d_imgs + b*frame_stride, `step` bytes per row.
 * d_mask: NULL, or masks at d_mask + b*mask_frame_stride (mask_frame_stride==0: one shared mask).
 * d_kps: B x cap records, d_desc: B x cap x 32 bytes, d_counts: B int32 (key points per frame;
 * a frame needing more than `cap` slots is truncated and flagged in plp_orb_last_batch_status).
 * Asynchronous on hip_stream. */
plp_status plp_orb_extract_batch_device(plp_orb* ctx, const uint8_t* d_imgs, int32_t B, int32_t rows,
                                        int32_t cols, size_t step, size_t frame_stride,
                                        const uint8_t* d_mask, size_t mask_step, size_t mask_frame_stride,
                                        plp_keypoint* d_kps, uint8_t* d_desc, int32_t cap,
                                        int32_t* d_counts, void* hip_stream);
/* Synchronises the last batch's stream and reports truncation / overflow (PLP_OK if none). */
plp_status plp_orb_last_batch_status(plp_orb* ctx);

/* Per-stage timing with HIP events recorded on the stream the kernels run on (profiling mode makes
 * every batch synchronous; leave it off in throughput runs).  ms7 = accumulated milliseconds of
 * {level-0 copy, pyramid, FAST cells, blur, quadtree, orientation+rBRIEF, whole batch}. */
plp_status plp_orb_set_profiling(plp_orb* ctx, int32_t enable);
plp_status plp_orb_get_stage_times(plp_orb* ctx, double* ms7, int64_t* n_batches);

/* orb_extractor::image_pyramid_ (public member, orb_extractor.h:101; read by match::stereo,
 * src/PLPSLAM/data/frame.cc:277-281).  Copies level `level` of frame `frame` of the last call to host. */
plp_status plp_orb_pyramid_level_size(const plp_orb* ctx, int32_t level, int32_t* rows, int32_t* cols);
plp_status plp_orb_pyramid_host(plp_orb* ctx, int32_t frame, int32_t level, uint8_t* dst, size_t dst_step);

/* match::stereo(...).compute(stereo_x_right, depths)  (src/PLPSLAM/match/stereo.cc:30-150, built in data/frame.cc:277-281):
 * per left key point the best right key point in its row band (Hamming < 75, octave +-1, disparity in [0, fx*b/b)),
 * 11x11 L1 patch slide on the pyramid level + parabola, 2x-median correlation rejection.  The image pyramids are the
 * ones `left` / `right` built in their last extract call (the reference passes orb_extractor::image_pyramid_).
 * Outputs: n_l floats each, -1 where no stereo match. */
plp_status plp_stereo_compute(plp_orb* left, plp_orb* right, const plp_keypoint* kps_l, int32_t n_l, const plp_keypoint* kps_r, int32_t n_r,
                              const uint8_t* desc_l, const uint8_t* desc_r, float focal_x_baseline, float true_baseline,
                              float* stereo_x_right, float* depths);
/* Batched: key points / descriptors / counts as produced by plp_orb_extract_batch_device of the two extractors. */
plp_status plp_stereo_compute_batch_device(plp_orb* left, plp_orb* right, const plp_keypoint* d_kps_l, const int32_t* d_cnt_l,
                                           const plp_keypoint* d_kps_r, const int32_t* d_cnt_r, const uint8_t* d_desc_l,
                                           const uint8_t* d_desc_r, int32_t cap, int32_t B, float focal_x_baseline, float true_baseline,
                                           float* d_x_right, float* d_depths, void* hip_stream);

/* Stage read-back for parity tests (synchronous; host destination).
 *   PLP_ORB_DBG_BLURRED   : the 7x7 sigma-2 blurred level image (orb_extractor.cc:148-149), rows x cols u8 dense
 *   PLP_ORB_DBG_CANDIDATES: keypts_to_distribute of a level (orb_extractor.cc:359,433) as int32 triples
 *                           (x, y, score) in border-relative coordinates, reference order
 *   PLP_ORB_DBG_SELECTED  : per-level quadtree output (orb_extractor.cc:443) as int32 triples (x, y, score)
 * *n_out = number of bytes (BLURRED) or triples written. */
typedef enum plp_orb_debug_id { PLP_ORB_DBG_BLURRED = 0, PLP_ORB_DBG_CANDIDATES = 1, PLP_ORB_DBG_SELECTED = 2 } plp_orb_debug_id;
plp_status plp_orb_debug_read(plp_orb* ctx, plp_orb_debug_id what, int32_t frame, int32_t level,
                              void* dst, size_t dst_bytes, int64_t* n_out);

/* ------------------------------------------------------------------------------------------
 * Line front-end — replaces feature::LineFeatureTracker (src/PLPSLAM/feature/line_extractor.h:61-104):
 * LSD detection (LSDDetectorC::detect with the options of line_extractor.cc:113-122, which wraps
 * cv::createLineSegmentDetector) + LBD description (BinaryDescriptor::compute) + the length filter and
 * the 2-D line functions of line_extractor.cc:134-159.
 * ---------------------------------------------------------------------------------------- */
typedef struct plp_keyline {   /* field-for-field cv::line_descriptor::KeyLine, 68 bytes (descriptor_custom.hpp:139-174) */
    float angle;
    int32_t class_id;
    int32_t octave;
    float pt_x, pt_y;
    float response;
    float size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength;
    int32_t numOfPixels;
} plp_keyline;

typedef struct plp_line plp_line;   /* one per LineFeatureTracker instance */
/* LineFeatureTracker(camera::base*): the camera only feeds the identity "undistortion" remap
 * (line_extractor.cc:40-86,103), which is elided; no parameter is needed. */
plp_status plp_line_create(int device, plp_line** out);
void plp_line_destroy(plp_line* ctx);

/* extract_LSD_LBD(img, frame_keylsd, frame_lbd_descr, keyline_functions)  (line_extractor.cc:88-160).
 * Host pointers, synchronous.  kl: cap records, lbd: cap x 32 bytes, linefn: cap x 3 doubles
 * (normalised (sx,sy,1) x (ex,ey,1)); the reference APPENDS to keyline_functions (:158) — the caller's
 * facade appends these rows.  *n_out = number of kept lines. */
plp_status plp_line_extract(plp_line* ctx, const uint8_t* img, int32_t rows, int32_t cols, size_t step,
                            plp_keyline* kl, uint8_t* lbd, double* linefn, int32_t cap, int32_t* n_out);
/* Batched replay form: B frames resident in HBM, results stay in HBM (B x cap records each). Asynchronous. */
plp_status plp_line_extract_batch_device(plp_line* ctx, const uint8_t* d_imgs, int32_t B, int32_t rows, int32_t cols,
                                         size_t step, size_t frame_stride, plp_keyline* d_kl, uint8_t* d_lbd,
                                         double* d_linefn, int32_t cap, int32_t* d_counts, void* hip_stream);
plp_status plp_line_last_batch_status(plp_line* ctx);
/* HIP-event stage timing (profiling mode makes batches synchronous).  ms9 = accumulated ms of {11-tap blur + x0.5 resize,
 * gradient + bins, seed order, region growing, key lines, 5-tap blur + Sobel, LBD, finalize, whole batch}. */
plp_status plp_line_set_profiling(plp_line* ctx, int32_t enable);
plp_status plp_line_get_stage_times(plp_line* ctx, double* ms9, int64_t* n_batches);
/* Tuning / parity tests: waves that share one frame in LSD region growing.  0 (default) = automatic: a workgroup of up to 8 waves per
 * frame (one sequential main wave, helpers that grow regions of later seeds speculatively and hand them over, k_lsd_grow_mw) for batches
 * of at most 256 frames -- the single-frame call of data/frame.cc:1146-1163 --, one wave per frame for larger batches; 1 = always one wave
 * per frame; 2..8 = that many waves for every batch of at most 256 frames.  The results are identical whichever is used. */
plp_status plp_line_set_grow_waves(plp_line* ctx, int32_t waves);
/* The order in which LSD visits its seed pixels.  OpenCV's lsd.cpp (reached from LSDDetector_custom.cpp:244-257) sorts every pixel by
 * gradient bin with std::sort and a comparator that looks at the bin only: inside a bin the order is whatever the C++ library's
 * (unstable) algorithm leaves, and region growing depends on it.
 *   PLP_SEED_ORDER_LIBSTDCXX  (default) the permutation libstdc++'s std::sort produces (introsort replayed on the device,
 *                             seed_sort_kernels.hip): bit-identical to a reference built with GCC's library
 *   PLP_SEED_ORDER_STABLE     bin descending, row-major inside a bin (what the LSD paper describes; cheaper: only defined pixels are
 *                             sorted, ~4.5 ms less per 2048 frames) -- for callers that do not need the reference's tie order
 * 3.5 % of the key lines differ between the two. */
typedef enum plp_seed_order { PLP_SEED_ORDER_STABLE = 0, PLP_SEED_ORDER_LIBSTDCXX = 1 } plp_seed_order;
plp_status plp_line_set_seed_order(plp_line* ctx, int32_t order);
plp_status plp_line_get_seed_order(const plp_line* ctx, int32_t* order);
/* Gives back device memory the current settings do not need: the exact seed order's buffers (630 KB per frame of the largest 640 x 480 batch seen) while
 * PLP_SEED_ORDER_STABLE is selected, and the several-waves grower's region lists.  plp_line_set_seed_order itself frees nothing (a caller may alternate the two
 * orders per batch at no cost).  Waits for the device (hipFree): call it where a pause is acceptable.  The caller's current device is left as it was. */
plp_status plp_line_trim(plp_line* ctx);

/* Stage read-back for parity tests (synchronous, host destination, frame of the last call):
 *   SCALED   u8 sh x sw dense (the 11-tap blur + x0.5 image LSD works on)
 *   ORDER    int32 seed order (pixel index y*sw + x) of the pixels whose level-line angle is defined (gradient magnitude > rho):
 *            region growing starts nowhere else, so only those are ordered
 *   RAW      float x 4 per LSD segment (x1,y1,x2,y2), in detection order
 *   ALL_KL   plp_keyline of every segment longer than min_length (before the >= 60 px filter)
 *   ALL_LBD  32 bytes per ALL_KL record
 *   SOBEL_DX / SOBEL_DY  int16 rows x cols */
typedef enum plp_line_debug_id { PLP_LINE_DBG_SCALED = 0, PLP_LINE_DBG_ORDER = 1, PLP_LINE_DBG_RAW = 2, PLP_LINE_DBG_ALL_KL = 3,
                                 PLP_LINE_DBG_ALL_LBD = 4, PLP_LINE_DBG_SOBEL_DX = 5, PLP_LINE_DBG_SOBEL_DY = 6,
                                 PLP_LINE_DBG_GROW_STATS = 7 /* int32[4]; one wave per frame: regions grown, pixels accepted, exact (in-band) decisions, 0;
                                    several waves per frame: regions the main wave grew itself, helper results taken, rejected, number of waves */ } plp_line_debug_id;
plp_status plp_line_debug_read(plp_line* ctx, plp_line_debug_id what, int32_t frame, void* dst, size_t dst_bytes, int64_t* n_out);
/* Host model of the bin ranking of match::angle_checker (the reference sorts its 30 histogram bins by size with std::sort,
 * src/PLPSLAM/match/angle_checker.h:165-176; the kernels reproduce libstdc++'s algorithm so that ties fall as in a reference built with
 * GCC): idx = the indices 0..n-1, n <= 64, in that order.  depth_limit < 0 = the library's recursion budget.  No GPU needed. */
int32_t plp_model_index_sort_host(const int32_t* sizes, int32_t n, int32_t depth_limit, uint32_t* idx);
/* Host model of the exact seed sort (csrc/seed_sort_model.hpp): std::__introsort_loop, in place, on n entries whose sort key is bits 20..29
 * (larger first), computed as the rank-paired partitions the kernel runs; depth_limit < 0 = the library's 2 * floor(log2 n).  skip_key > 0:
 * parts that can only hold keys below it are left as they are, as the kernel leaves the undefined pixels of a frame (they are sorted along
 * but never seed a region); the order of the entries with keys >= skip_key after a stable sort by key is std::sort's all the same.
 * No GPU needed.  Returns 0, or -1 for a bad argument. */
int32_t plp_model_seed_introsort_host(uint32_t* entries, int64_t n, int32_t depth_limit, uint32_t skip_key);
/* Test entry: the KERNEL's introsort loop on caller-made entries (host pointer, in place), one workgroup, chosen recursion budget and skip key.
 * variant 0: the kernel configuration of large batches (4 waves, 4096-entry LDS window), 1: of batches up to 256 frames (16 waves, 24576 entries).
 * n_live (may be NULL): the length of the array's LIVE part.  With a skip key the kernel stops storing into the right part of a global-memory partition
 * whose pivot key lies below it once that part is the end of the live array (only keys below the skip key live there, nothing reads them again): entries
 * [0, n_live) equal the host model's, entries behind are unspecified (stale copies).  Without a skip key n_live = n. */
plp_status plp_seed_introsort_debug(int32_t device, uint32_t* entries, int64_t n, int32_t depth_limit, uint32_t skip_key, int32_t variant, int32_t* n_live);
/* Host model of the LSD gradient kernel's (float)cos((double)a), (float)sin((double)a) fast path (csrc/sincos_ziv.hpp): proven[i] = 0 marks the
 * arguments for which the kernel falls back to the general f64 routine.  Returns the number of proven arguments.  No GPU needed. */
int32_t plp_model_sincos_host(const float* a, int64_t n, float* c, float* s, uint8_t* proven);
plp_status plp_line_scaled_size(const plp_line* ctx, int32_t* rows, int32_t* cols);
/* Diagnostics of frame 0 of the last batch, 12 values.  One wave per frame: shader cycles {whole wave, region_grow, region2rect, refine},
 * regions grown, pixels grown, 0 x 6.  Several waves per frame, the main wave's view: cycles {whole, waiting for helpers, growing regions
 * itself}, helper attempts | give-ups << 32, regions it grew itself, results taken | rejected << 32, cycles {taking results, publishing its
 * own regions, group set-up}, 0 x 3. */
plp_status plp_line_debug_grow_profile(plp_line* ctx, int64_t* out12);

/* ------------------------------------------------------------------------------------------
 * Hamming matchers, array form — replace the inner loops of the reference's src/PLPSLAM/match directory.
 *
 * The reference's matchers walk data::frame / data::landmark objects (match/projection.h,
 * match/robust.h); the host facade flattens what those loops read into plain arrays, calls one of
 * the entry points below and writes the returned associations back.  Targets = key points of the
 * current frame, queries = landmarks / last-frame key points / key-frame key points IN THE
 * REFERENCE'S ITERATION ORDER (results depend on it: an accepted query blocks its key point for all
 * later queries).  B independent problems per call; arrays are B x n_cap / B x m_cap, row-major.
 * ---------------------------------------------------------------------------------------- */
typedef struct plp_keyline plp_keyline_fwd_;
typedef struct plp_matcher plp_matcher;   /* owns a HIP stream and scratch; one per calling thread */
plp_status plp_matcher_create(int device, plp_matcher** out);
void plp_matcher_destroy(plp_matcher* ctx);

typedef enum plp_match_mode {
    PLP_MATCH_MODE_LANDMARKS = 0,   /* projection::match_frame_and_landmarks          match/projection.cc:37-121  */
    PLP_MATCH_MODE_LAST_FRAME = 1,  /* projection::match_current_and_last_frames      match/projection.cc:214-358 */
    PLP_MATCH_MODE_BRUTE_FORCE = 2, /* robust::brute_force_match                      match/robust.cc:257-385     */
    PLP_MATCH_MODE_LANDMARKS_LINE = 3,  /* projection::match_frame_and_landmarks_line      match/projection.cc:124-212 */
    PLP_MATCH_MODE_LAST_FRAME_LINE = 4, /* projection::match_current_and_last_frames_line  match/projection.cc:361-527 */
    PLP_MATCH_MODE_BOW = 5,         /* bow_tree::match_frame_and_keyframe / match_keyframes   match/bow_tree.cc:41-165, 167-307 */
    PLP_MATCH_MODE_FUSE = 6,        /* fuse::replace_duplication, the search part              match/fuse.cc:169-298        */
    PLP_MATCH_MODE_FUSE_LINE = 7,   /* fuse::replace_duplication_line, the search part         match/fuse.cc:335-470        */
    PLP_MATCH_MODE_TRIANGULATION = 8/* robust::match_for_triangulation                         match/robust.cc:43-216       */
} plp_match_mode;

/* plp_match_args.flags */
#define PLP_MATCH_FLAG_NO_CHI2 1        /* FUSE: no chi-square gates (fuse::detect_duplication fuse.cc:40-166,
                                           projection::match_keyframes_mutually projection.cc:894-1142) */
#define PLP_MATCH_FLAG_SIGNED_LEVEL 2   /* FUSE: octave window [q_level-1, q_level] in SIGNED arithmetic (detect_duplication
                                           declares `const int pred_scale_level`, fuse.cc:109,126) */
#define PLP_MATCH_FLAG_UNSIGNED_LEVEL 4 /* LAST_FRAME with level_window 1: the manual window of match_by_Sim3_transform
                                           (projection.cc:862) in unsigned arithmetic, q_level == 0 matches nothing */
#define PLP_MATCH_FLAG_MARK_INVALIDATED 8 /* out_match = -2 (instead of -1) for a key point whose match the orientation check
                                           removed: the reference writes nullptr there (projection.cc:350-354), which differs
                                           from "never matched" when the slot held an observation-less landmark before */
/* plp_match_args.level_window (LAST_FRAME / LAST_FRAME_LINE): 0 = from `direction`, 1 = [q_level-1, q_level],
 * 2 = [q_level-1, q_level+1] */

typedef struct plp_match_grid {     /* camera::base grid (camera/base.h:91) used by data::get_keypoints_in_cell */
    float min_x, min_y;             /* img_bounds_.min_x_/min_y_ (float, camera/base.h:78-81) */
    double inv_cell_width, inv_cell_height;   /* double, camera/base.h:158-160 */
    int32_t cols, rows;             /* 64 x 48 */
} plp_match_grid;

typedef struct plp_match_args {
    int32_t mode;                   /* plp_match_mode */
    int32_t B, n_cap, m_cap;        /* B > 0; n_cap = 0 (a frame without key points / key lines) or m_cap = 0 (no landmarks / queries) is a valid call: nothing can match --
                                     * the reference's loops do not run -- every out_match slot becomes -1, every out_num 0 (fuse modes: out_query_best -1); input pointers
                                     * of the empty side may be NULL */
    /* targets (current frame): data::frame::undist_keypts_, descriptors_, stereo_x_right_, and
     * "landmarks_[idx] && landmarks_[idx]->has_observation()" as a byte flag */
    const plp_keypoint* t_kps;      /* B x n_cap (ignored in brute-force mode) */
    const uint8_t* t_desc;          /* B x n_cap x 32 */
    const float* t_x_right;         /* B x n_cap or NULL */
    const uint8_t* t_occupied;      /* B x n_cap or NULL */
    const float* t_angle;           /* brute-force mode: B x n_cap key point angles of frame 1 */
    const int32_t* t_counts;        /* B, or NULL = n_cap everywhere */
    /* queries */
    const uint8_t* q_valid;         /* B x m_cap or NULL: the skip tests at the top of the reference loops */
    const float* q_reproj;          /* B x m_cap x 2: reproj_in_tracking_ / reprojected last-frame landmark */
    const float* q_x_right;         /* B x m_cap or NULL */
    const int32_t* q_level;         /* B x m_cap: scale_level_in_tracking_ / last_frm.keypts_[i].octave */
    const float* q_angle;           /* B x m_cap: needed when check_orientation */
    const uint8_t* q_desc;          /* B x m_cap x 32 */
    const uint8_t* q_has_obs;       /* B x m_cap or NULL (= all 1): landmark::has_observation() */
    const int32_t* q_counts;        /* B, or NULL */
    float margin, lowe_ratio;       /* match::base(lowe_ratio, check_orientation) */
    int32_t direction;              /* LAST_FRAME: 0 neither, 1 assume_forward, 2 assume_backward */
    int32_t check_orientation;
    int32_t num_levels;
    const float* scale_factors;     /* HOST pointer, num_levels floats (frame::scale_factors_) */
    plp_match_grid grid;
    /* line modes (targets = key lines of the current frame, candidates by data::get_keylines_in_cell,
     * data/common.cc:315-363): t_kl replaces t_kps, t_desc = _lbd_descr, q_reproj/q_reproj2 = reprojected start /
     * end point, q_desc = Line::get_descriptor().  t_kp_octave[i] = undist_keypts_.at(i).octave, the key POINT
     * octave the reference reads with a LINE index (projection.cc:187,192; LANDMARKS_LINE only).  RGB-D stereo
     * gate of LAST_FRAME_LINE: t_x_right / t_x_right2 = _stereo_x_right_cooresponding_to_keylines first / second,
     * q_x_right / q_x_right2 = x_right of the reprojected start / end point, is_rgbd != 0. */
    const plp_keyline* t_kl;        /* B x n_cap */
    const int32_t* t_kp_octave;     /* B x n_cap */
    const float* t_x_right2;        /* B x n_cap or NULL */
    const float* q_reproj2;         /* B x m_cap x 2 */
    const float* q_x_right2;        /* B x m_cap or NULL */
    int32_t is_rgbd;
    int32_t num_levels_lsd;         /* last_frm._num_scale_levels_lsd (upper bound of the assume_forward window) */
    /* BOW mode: queries = key-frame features listed in BoW node order (the order the reference walks the feature
     * vector in), q_group / t_group = node id of every feature (frame features of one node are visited in ascending
     * index); q_valid = feature has a live landmark; t_occupied = static skip of a target (match_keyframes: key frame 2
     * feature without a live landmark); q_angle / t_angle when check_orientation.  Accept: best <= 50 and
     * lowe_ratio * second >= best.
     * FUSE mode: independent queries (no blocking): window margin * scale_factors[q_level] around q_reproj_d (all
     * octaves), then octave in [q_level - 1, q_level] WITH THE REFERENCE'S UNSIGNED ARITHMETIC (q_level == 0 rejects
     * everything, fuse.cc:236), the chi-square gates on the f64 reprojection (5.99146 / 7.81473 with
     * inv_level_sigma_sq, stereo when t_x_right >= 0), best <= 50.  Result per query in out_query_best. */
    const int32_t* q_group;         /* B x m_cap */
    const int32_t* t_group;         /* B x n_cap */
    const double* q_reproj_d;       /* B x m_cap x 2 (FUSE) */
    const float* inv_level_sigma_sq;/* HOST pointer, num_levels floats (FUSE) */
    int32_t* out_query_best;        /* B x m_cap (FUSE, FUSE_LINE): best key point / key line per query, -1 = none */
    /* Variants of the same loops:
     *  - projection::match_frame_and_keyframe[_line] (projection.cc:529-645, 648-779) = LAST_FRAME[_LINE] with
     *    direction 0, t_x_right NULL / is_rgbd 0, t_occupied = "landmarks_[i] != nullptr", q_has_obs NULL and
     *    hamm_dist_thr = the caller's threshold;
     *  - projection::match_by_Sim3_transform (:781-892) = LAST_FRAME, level_window 1, PLP_MATCH_FLAG_UNSIGNED_LEVEL,
     *    hamm_dist_thr 50, check_orientation 0, t_occupied = "matched_lms_in_keyfrm[i] != nullptr";
     *  - projection::match_keyframes_mutually (:894-1142) = two FUSE calls with PLP_MATCH_FLAG_NO_CHI2 and
     *    hamm_dist_thr 100, then the cross check (:1124-1139) on the two out_query_best arrays;
     *  - fuse::detect_duplication (fuse.cc:40-166) = FUSE with NO_CHI2 | SIGNED_LEVEL.
     * FUSE_LINE: targets t_kl / t_desc (LBD), queries q_reproj_d = reprojected start point, q_reproj2_d = end point
     *  (f64), window = data::get_keylines_in_cell(margin * scale_factors[q_level]) without octave test, gate
     *  5.99146 < (e_sp^2 + e_ep^2) * inv_level_sigma_sq[keyline.octave] in f64, best <= 50 (fuse.cc:431-470).
     * TRIANGULATION: BOW-style node-guided search between two key frames; q = features of key frame 1 in BoW node
     *  order (q_valid = "has no landmark"), t = features of key frame 2 (t_occupied = "has a landmark"); q_x_right /
     *  t_x_right >= 0 mark stereo key points; q_level = undist_keypts_[idx_1].octave; gates in f64 on the bearings:
     *  epipole test (both mono: skip if 0.99862953475 < epipole . bearing_2) and check_epipolar_constraint
     *  (robust.cc:387-405); Hamming <= 50, equal distances: the LATER candidate wins (`best < dist` skips, :124);
     *  exclusive on t; angle check on q_angle - t_angle.  `epipolar` = per problem 12 doubles: E_12 row-major, then
     *  the epipole bearing in key frame 2. */
    int32_t hamm_dist_thr;          /* 0 = the mode's own threshold */
    int32_t level_window;
    int32_t flags;
    const double* q_reproj2_d;      /* B x m_cap x 2 (FUSE_LINE) */
    const double* q_bearing;        /* B x m_cap x 3 (TRIANGULATION) */
    const double* t_bearing;        /* B x n_cap x 3 (TRIANGULATION) */
    const double* epipolar;         /* B x 12 (TRIANGULATION) */
    /* outputs: out_match[b][t] = index of the query associated with key point t (-1 = none),
     * out_num[b] = the matcher's return value (num_matches) */
    int32_t* out_match;             /* B x n_cap */
    int32_t* out_num;               /* B */
    /* Rows between two problems' blocks in q_desc (0 = m_cap).  In a batched replay the queries of frame b are the features of the
     * frames before it: with q_desc_stride = the per-frame capacity, q_desc may point INTO the batch's own descriptor array (one
     * frame back, m_cap = cap; or two frames back, m_cap = 2 cap: overlapping windows) and no descriptor is copied. */
    int32_t q_desc_stride;
    /* Performance hint for the windowed point modes (LANDMARKS, LAST_FRAME), 0 = none: an upper bound the caller EXPECTS for t_counts[b]
     * (a tracker knows its extractor's max_num_keypoints; the arrays are usually strided by a larger capacity n_cap).  The matcher then
     * keeps only that many targets of a frame in LDS -- more of its workgroups fit a compute unit -- and reads the targets of a frame that
     * has more from memory: results never depend on the hint. */
    int32_t t_count_hint;
} plp_match_args;

/* All array pointers in `a` are DEVICE pointers (except scale_factors); asynchronous on hip_stream. */
plp_status plp_match_device(plp_matcher* ctx, const plp_match_args* a, void* hip_stream);
/* Same with HOST pointers for one call (B problems are staged to HBM and back); synchronous. */
plp_status plp_match_host(plp_matcher* ctx, const plp_match_args* a);

/* Batched replay (bench.py, replay_driver.py): the queries of the tracker's per-frame matcher calls, built on the
 * device from the features of the preceding frames of the batch -- what tracking_module does on the host with poses, here for a
 * camera that pans by (shift_x, shift_y) pixels per frame (example/run_tum_rgbd_slam_with_line.cc replayed without a map).
 * feat_*: [(halo + B)][cap] rows; row halo + b is frame b of this rank's block, rows 0 .. halo-1 the predecessor's tail (halo >= 2).
 *   last-frame queries  (frame b-1 moved by 1 x shift):  q1_reproj [B][cap][2], q1_level [B][cap], q1_angle [B][cap], q1_counts [B]
 *   local-landmark queries (frames b-2, b-1 moved by 2 x / 1 x shift): q2_reproj [B][2 cap][2], q2_level [B][2 cap], q2_valid [B][2 cap]
 * Descriptors are not copied: pass q_desc = feat_desc + (halo - 1) * cap * 32 (resp. halo - 2) with q_desc_stride = cap. */
plp_status plp_replay_point_queries_device(const plp_keypoint* feat_kps, const int32_t* feat_counts, int32_t halo, int32_t B, int32_t cap, float shift_x,
                                           float shift_y, float* q1_reproj, int32_t* q1_level, float* q1_angle, int32_t* q1_counts, float* q2_reproj,
                                           int32_t* q2_level, uint8_t* q2_valid, void* hip_stream);
/* key lines of frame b-1, both end points moved by the shift: q_sp / q_ep [B][cap][2], q_level [B][cap] (KeyLine::octave), q_counts [B].
 * Optional (all four or none, halo >= 2): the local LINE landmarks of frame b for projection::match_frame_and_landmarks_line
 * (match/projection.cc:124-212, called per frame from tracking_module.cc:1060) = key lines of frame b-2 moved by 2 x shift, then those of
 * frame b-1 moved by 1 x shift: q2_sp / q2_ep [B][2 cap][2], q2_level [B][2 cap], q2_valid [B][2 cap]; their LBD rows are read in place
 * (q_desc = feat_lbd + (halo - 2) * cap * 32, q_desc_stride = cap).
 * Optional: t_kp_octave [B][cap] = frame b's undist_keypts_[i].octave for i < cap (the key POINT octave that matcher reads with a LINE
 * index, projection.cc:187,192) from feat_kps [halo + B][kp_cap] / feat_kp_counts [halo + B]. */
plp_status plp_replay_line_queries_device(const plp_keyline* feat_kl, const int32_t* feat_counts, int32_t halo, int32_t B, int32_t cap, float shift_x,
                                          float shift_y, float* q_sp, float* q_ep, int32_t* q_level, int32_t* q_counts, float* q2_sp, float* q2_ep,
                                          int32_t* q2_level, uint8_t* q2_valid, const plp_keypoint* feat_kps, const int32_t* feat_kp_counts, int32_t kp_cap,
                                          int32_t* t_kp_octave, void* hip_stream);

/* Host boundary of the batched replay: the live rows of a padded per-frame array packed back to back, so that the device-to-host copy
 * moves the features that exist instead of the capacity they were allotted (the std::vector<cv::KeyPoint> / cv::Mat rows that
 * orb_extractor::extract and extract_LSD_LBD hand back hold exactly that many entries, feature/orb_extractor.cc:124-132).
 * dst[offsets[b] + i] = src[b][i] for i < min(counts[b], cap), rows of row_bytes bytes (a multiple of 4); offsets [B + 1] = exclusive
 * prefix sum of the clamped counts, offsets[B] = total rows.  compute_offsets != 0 computes them first; pass 0 to reuse the offsets of an
 * earlier call with the same counts (key points + descriptors + matches share one set).  Asynchronous on hip_stream. */
plp_status plp_pack_rows_device(const void* src, const int32_t* counts, int32_t B, int32_t cap, int32_t row_bytes, void* dst, int64_t* offsets,
                                int32_t compute_offsets, void* hip_stream);

/* area::match_in_consistent_area(frm_1, frm_2, prev_matched_pts, matched_indices_2_in_frm_1, margin)
 * (src/PLPSLAM/match/area.cc:33-153; monocular initialisation, module/initializer.cc:191-192).  Host pointers, one
 * problem, synchronous.  kps_1/kps_2 = undist_keypts_ of the two frames, prev_matched_pts = n1 x 2 floats (updated in
 * place for matched key points), matched_2_in_1 = n1 int32 (idx_2 or -1); *num_matches = return value. */
plp_status plp_match_area_host(plp_matcher* ctx, const plp_keypoint* kps_1, const uint8_t* desc_1, int32_t n1, const plp_keypoint* kps_2,
                               const uint8_t* desc_2, int32_t n2, const plp_match_grid* grid, float* prev_matched_pts, int32_t margin,
                               float lowe_ratio, int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches);

/* cv::line_descriptor::BinaryDescriptorMatcher::match(query, train, matches) — exact 1-NN over LBD descriptors by
 * multi-index hashing (src/PLPSLAM/feature/line_descriptor/binary_descriptor_matcher.cpp:197-255, 597-818), used for the
 * stereo line association (data/frame.cc:496-533) and two-key-frame line triangulation (mapping_module.cc:481-531).
 * train_idx[q] = DMatch.trainIdx (among equally near train lines: the one MIH discovers first), dist[q] = DMatch.distance.
 * Nothing within Hamming distance 128 -> (-1, 256) (the reference reads uninitialised memory there). */
plp_status plp_lbd_match_1nn_host(plp_matcher* ctx, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t* train_idx, int32_t* dist);
/* B problems: q is B x nq_cap x 32, t is B x nt_cap x 32, counts per problem (NULL = cap); asynchronous. */
plp_status plp_lbd_match_1nn_device(plp_matcher* ctx, const uint8_t* d_q, const int32_t* d_q_counts, int32_t nq_cap, const uint8_t* d_t,
                                    const int32_t* d_t_counts, int32_t nt_cap, int32_t B, int32_t* d_train_idx, int32_t* d_dist, void* hip_stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Post-extract per-key-point step of every data::frame constructor (src/PLPSLAM/data/frame.cc:68-86, 110-128, ...;
 * SURVEY.md 8(f) item 1), so that features can stay in HBM between extraction and matching:
 *   camera::perspective::undistort_keypoints            camera/perspective.cc:130-162 (cv::undistortPoints, EPS|MAX_ITER 20 1e-6,
 *                                                        float camera matrix / distortion vector as the reference stores them)
 *   camera::perspective::convert_keypoints_to_bearings  camera/perspective.cc:165-175
 *   data::frame::compute_stereo_from_depth              data/frame.cc:1169-1219 (key points, and key lines when given)
 * (data::assign_keypoints_to_grid, data/common.cc:205-231, is what the matcher's preparation kernel builds from undist.)
 * undist gets pt / angle / size / octave of the distorted key point, response 0 and class_id -1, as the reference's resize()
 * + field copies do.  depth: B x rows x cols f32 (depth_step = bytes per row) or NULL (then x_right / depths may be NULL);
 * key-line outputs are written only for lines with both end-point depths >= 0, like the reference (pre-fill them).
 * Device pointers, asynchronous on hip_stream. */
typedef struct plp_camera {
    double fx, fy, cx, cy;          /* camera::perspective fx_, fy_, cx_, cy_ */
    double k1, k2, p1, p2, k3;      /* distortion */
    double focal_x_baseline;        /* camera::base focal_x_baseline_ */
} plp_camera;
plp_status plp_post_extract_device(plp_matcher* ctx, const plp_camera* cam, const plp_keypoint* d_kps, const int32_t* d_counts, int32_t cap,
                                   int32_t B, const float* d_depth, int32_t rows, int32_t cols, size_t depth_step, size_t depth_frame_stride,
                                   plp_keypoint* d_undist, double* d_bearings, float* d_x_right, float* d_depths,
                                   const plp_keyline* d_kl, const int32_t* d_kl_counts, int32_t kl_cap, float* d_kl_depths,
                                   float* d_kl_x_right, void* hip_stream);
/* One frame, host pointers, synchronous (NULL for the parts that are not wanted, as above). */
plp_status plp_post_extract_host(plp_matcher* ctx, const plp_camera* cam, const plp_keypoint* kps, int32_t n, const float* depth, int32_t rows,
                                 int32_t cols, size_t depth_step, plp_keypoint* undist, double* bearings, float* x_right, float* depths,
                                 const plp_keyline* kl, int32_t n_kl, float* kl_depths, float* kl_x_right);

/* Input side: util::convert_to_grayscale (src/PLPSLAM/util/image_converter.cc:33-75, cv::cvtColor
 * RGB/BGR[A] -> gray on CV_8U) and util::convert_to_true_depth (:77-80, convertTo(CV_32F, 1 / depthmap_factor)), so that the
 * raw colour / 16-bit depth frames can go straight to HBM.  B frames, device pointers, asynchronous.
 * channels 3 or 4; color_order 0 = RGB(A), 1 = BGR(A).  src_is_u16 != 0: CV_16UC1 depth, else CV_32FC1. */
plp_status plp_convert_to_grayscale_device(plp_matcher* ctx, const uint8_t* d_src, int32_t rows, int32_t cols, size_t src_step,
                                           size_t src_frame_stride, int32_t channels, int32_t color_order, int32_t B, uint8_t* d_gray,
                                           size_t gray_step, size_t gray_frame_stride, void* hip_stream);
plp_status plp_convert_to_true_depth_device(plp_matcher* ctx, const void* d_src, int32_t src_is_u16, int32_t rows, int32_t cols, size_t src_step,
                                            size_t src_frame_stride, double depthmap_factor, int32_t B, float* d_dst, size_t dst_step,
                                            size_t dst_frame_stride, void* hip_stream);

/* util::stereo_rectifier (src/PLPSLAM/util/stereo_rectifier.cc:38-85), perspective model.
 * plp_rectify_map_device = the constructor's cv::initUndistortRectifyMap(K, D, R, K_rect, img_size, CV_32F, map_x, map_y)
 * for one eye (:61-62): K, R row-major 3x3 doubles and D (n_dist in {0, 4, 5, 8, 12}: k1 k2 p1 p2 [k3 [k4 k5 k6 [s1..s4]]])
 * as read from the yaml; rect_cam = the rectified camera, whose fx, fy, cx, cy are rounded to float like
 * camera::perspective::cv_cam_matrix_ (perspective.cc:47).  Writes rows x cols CV_32F maps (map_step bytes per row).
 * plp_remap_linear_device = rectify()'s cv::remap(in, out, map_x, map_y, cv::INTER_LINEAR) (:83-84) on B 8UC1 frames that
 * share one map pair: 1/32-pixel fixed point, 15-bit weights, constant border 0.  Device pointers, asynchronous. */
plp_status plp_rectify_map_device(plp_matcher* ctx, const double* K, const double* D, int32_t n_dist, const double* R,
                                  const plp_camera* rect_cam, int32_t rows, int32_t cols, float* d_map_x, float* d_map_y, size_t map_step,
                                  void* hip_stream);
/* The "fisheye" StereoRectifier.model (util/stereo_rectifier.cc:65-70, the TUM-VI yaml): cv::fisheye::initUndistortRectifyMap,
 * equidistant model with 4 coefficients.  Not bit-defined like the perspective map: OpenCV inverts K_rect * R by SVD (here
 * closed form) and the map goes through atan(); the tests hold it to 1e-4 px against the oracle.  The remap is the same. */
plp_status plp_rectify_map_fisheye_device(plp_matcher* ctx, const double* K, const double* D4, const double* R, const plp_camera* rect_cam,
                                          int32_t rows, int32_t cols, float* d_map_x, float* d_map_y, size_t map_step, void* hip_stream);
plp_status plp_remap_linear_device(plp_matcher* ctx, const uint8_t* d_src, int32_t rows, int32_t cols, size_t src_step, size_t src_frame_stride,
                                   const float* d_map_x, const float* d_map_y, size_t map_step, int32_t dst_rows, int32_t dst_cols, int32_t B,
                                   uint8_t* d_dst, size_t dst_step, size_t dst_frame_stride, void* hip_stream);

/* Planar_Mapping_module::create_ColorToPlane (src/PLPSLAM/planar_mapping_module.cc:185-345), the per-key-point part
 * (BASELINE config 5): labels[b][i] = colour label c0 + (c1 << 8) + (c2 << 16) of the CV_8UC3
 * instance mask under undistorted key point i, or 0 when the point is flagged invalid (valid == NULL: all valid), outside
 * the mask, on label 0, or (check_3x3_window) not surrounded by its own label.  The `> 0` neighbour tests of the
 * reference are kept (row 0 / column 0 are never looked at).  New data::Plane objects and add_landmark stay on the host.
 * Device pointers, asynchronous. */
plp_status plp_color_vote_device(plp_matcher* ctx, const uint8_t* d_mask, int32_t rows, int32_t cols, size_t mask_step,
                                 size_t mask_frame_stride, const plp_keypoint* d_undist, const uint8_t* d_valid, const int32_t* d_counts,
                                 int32_t cap, int32_t B, int32_t check_3x3_window, int32_t* d_labels, void* hip_stream);

/* Bag-of-words transform: data::frame::compute_bow / data::keyframe::compute_bow
 * (src/PLPSLAM/data/frame.cc:785-795) = bow_vocab_->transform(to_desc_vec(descriptors_), bow_vec_, bow_feat_vec_, 4) of
 * DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (data/bow_vocabulary.h:22; the USE_DBOW2 build).  DBoW2 and the
 * vocabulary file are not part of the reference tree, so the tree is handed over flat (the host walks m_nodes once after
 * loadFromBinaryFile): node 0 is the root, children of node i are children[child_offset[i] .. child_offset[i+1]) in
 * the order of Node::children (ties go to the first), leaves carry word_id and weight.
 *   accumulate: 1 for TF / TF_IDF weighting (a word's weight is added once per feature), 0 for IDF / BINARY;
 *   norm: what ScoringObject::mustNormalize asks for: 0 none (DOT_PRODUCT), 1 L1 (L1_NORM, CHI_SQUARE, KL, BHATTACHARYYA), 2 L2.
 * Outputs per frame b (arrays [B][cap]):
 *   word_id / node_id  per feature (may be NULL): the word, and the node at level L - levelsup that FeatureVector files
 *                      the feature under; 0xFFFFFFFF for a stopped word (weight <= 0), which enters neither map.  node_id
 *                      is what PLP_MATCH_MODE_BOW takes as q_group / t_group.
 *   bow_word / bow_value / n_bow   the BowVector (std::map<WordId, WordValue>) in key order, normalised;
 *   fv_node / fv_feat / n_fv       the FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened in key order,
 *                                  the features of a node in increasing index.
 * Limits: cap <= 8192 (4096 until round 6: the per-frame maps are sorted in LDS, 16 bytes per descriptor).  One transform in flight per vocabulary handle.  Device pointers, asynchronous. */
typedef struct plp_bow_vocab plp_bow_vocab;
typedef struct plp_bow_tree {
    int32_t n_nodes;               /* >= 2 */
    int32_t L;                     /* depth levels (m_L), 6 for the ORB vocabulary */
    const int32_t* child_offset;   /* n_nodes + 1 */
    const int32_t* children;       /* n_nodes - 1 */
    const uint8_t* node_desc;      /* n_nodes x 32 (row 0, the root, is not read) */
    const double* node_weight;     /* n_nodes (read at leaves) */
    const uint32_t* node_word;     /* n_nodes (read at leaves) */
    int32_t accumulate;
    int32_t norm;
} plp_bow_tree;
plp_status plp_bow_vocab_create(int device, const plp_bow_tree* tree, plp_bow_vocab** out);   /* host pointers, copied */
void plp_bow_vocab_destroy(plp_bow_vocab* vocab);
plp_status plp_bow_transform_device(plp_bow_vocab* vocab, const uint8_t* d_desc, const int32_t* d_counts, int32_t cap, int32_t B, int32_t levelsup,
                                    uint32_t* d_word_id, uint32_t* d_node_id, uint32_t* d_bow_word, double* d_bow_value, int32_t* d_n_bow,
                                    uint32_t* d_fv_node, uint32_t* d_fv_feat, int32_t* d_n_fv, void* hip_stream);
/* One frame, host pointers, synchronous; the output arrays hold n entries. */
plp_status plp_bow_transform_host(plp_bow_vocab* vocab, const uint8_t* desc, int32_t n, int32_t levelsup, uint32_t* word_id, uint32_t* node_id,
                                  uint32_t* bow_word, double* bow_value, int32_t* n_bow, uint32_t* fv_node, uint32_t* fv_feat, int32_t* n_fv);

/* landmark::compute_descriptor (src/PLPSLAM/data/landmark.cc:181-245) and Line::compute_descriptor
 * (data/landmark_line.cc:256-320), the search part, for L landmarks at once: landmark l owns the
 * descriptors descs[offsets[l] .. offsets[l+1]) (32 B rows, observation order); best_idx[l] = the row (relative to
 * offsets[l]) whose median Hamming distance to all rows of the landmark -- itself included, rank (unsigned)(0.5 * (n - 1)) --
 * is smallest, first such row; -1 for a landmark without rows.  At most 1024 rows per landmark. */
plp_status plp_landmark_descriptor_device(plp_matcher* ctx, const uint8_t* d_descs, const int32_t* d_offsets, int32_t L, int32_t* d_best_idx,
                                          void* hip_stream);
plp_status plp_landmark_descriptor_host(plp_matcher* ctx, const uint8_t* descs, const int32_t* offsets, int32_t L, int32_t* best_idx);

/* Diagnostics: {exact full rescans, resolve rounds, 0, 0} accumulated over all calls of this context (synchronous). */
plp_status plp_match_debug_counters(plp_matcher* ctx, int64_t* out4);

/* compute_descriptor_distance_32 over all pairs (match/base.h:43-68): dist[q*nt + t], u16.
 * Device pointers, asynchronous.  (K16: input of brute-force style matchers on the host side.) */
plp_status plp_hamming_matrix_device(plp_matcher* ctx, const uint8_t* d_q, int32_t nq, const uint8_t* d_t, int32_t nt,
                                     uint16_t* d_dist, void* hip_stream);
plp_status plp_hamming_matrix_host(plp_matcher* ctx, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, uint16_t* dist);

#ifdef __cplusplus
}
#endif
#endif /* PLP_FRONT_H */
