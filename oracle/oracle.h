/* oracle.h — CPU restatement of the reference algorithms on the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vpp_amd/ (the product) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker / the timed
 * CPU baseline.  Every function cites the reference file:line it restates (paths relative to the
 * reference tree, matt-42/vpp branch phd_work).  Same descriptor struct as the C ABI, but HOST pointers.
 *
 * Parity pinning: see oracle/README.md — the restatement is checked against (a) the reference's own
 * headers compiled unmodified against shims (oracle/ref -> oracle/_ref/libvpp_ref.so, built only where
 * /root/reference exists), (b) golden vectors generated from that build and committed under tests/golden/,
 * (c) the reference's known-answer tests (tests/pyrlk.cc, benchmarks' inline checkers).
 */
#ifndef VPP_ORACLE_H_
#define VPP_ORACLE_H_
#include "../include/vpp_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

int orc_num_threads(void); /* 1 for the parity build; omp_get_max_threads() for the timing build */

int orc_pixelwise_binary(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b);
int orc_copy(const vpp_image_desc* dst, const vpp_image_desc* src, int with_border);
int orc_fill(const vpp_image_desc* img, const void* value, int with_border);
int orc_box_filter(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C);
int orc_fill_border(const vpp_image_desc* img, int mode, const void* value);
int orc_lowpass5(const vpp_image_desc* out, const vpp_image_desc* in);
int orc_pyr_down(const vpp_image_desc* next, const vpp_image_desc* prev);
int orc_scharr(const vpp_image_desc* out, const vpp_image_desc* in);
int orc_fast9_detect(const vpp_image_desc* src, int th, const vpp_image_desc* mask, int mode, int block_size,
                     int compat, int32_t* out_rc, int32_t* out_scores, int capacity, int* count);
int orc_fast9_scores(const vpp_image_desc* src, int th, const int32_t* rc, int n, int32_t* out_scores);
int orc_pyrlk_match(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nlevels,
                    vpp_keypoint_f32* kps, int n, int winsize, float min_ev, float max_err, int max_iterations,
                    float convergence_delta, int min_scale, float* out_dist);
int orc_lucas_kanade(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nlevels,
                     const float* pts, const float* prediction, int n, int winsize, int min_ev, int niterations,
                     int delta, float* out_flow, float* out_dist);
int orc_semi_dense_optical_flow(const vpp_image_desc* i1, const vpp_image_desc* i2, const int32_t* kps, int n,
                                int winsize, int nscales, int min_scale, int propagation, int patchsize,
                                int32_t* out_pos, int32_t* out_dist, uint8_t* out_valid);
/* scalar FAST-9 segment test on the true ring (fast.hpp:80-112, :25-34): returns 0/1 */
int orc_is_fast9_keypoint(const vpp_image_desc* src, int r, int c, int th);
/* imageNd::linear_interpolate (imageNd.hpp:280-300) on one point; out has `channels` floats holding the
 * value AFTER the cast back to V (so integers for integer V). */
int orc_linear_interpolate(const vpp_image_desc* img, float pr, float pc, float* out);

/* colorspace_conversions.hh:10-33,36-48 (+ the ingest chain of examples/video_extruder.cc:46-48 when mirror != 0) */
int orc_rgb_to_graylevel(const vpp_image_desc* dst, const vpp_image_desc* src, int mirror);
/* video_extruder/video_extruder.hpp:95-110 (rc: n host (row, col) pairs) */
int orc_keypoint_mask(const vpp_image_desc* mask, const int32_t* rc, int n, int spacing);
int orc_keypoint_merge(const int32_t* pos_rc, const int32_t* age, int n, int nrows, int ncols, int spacing, uint8_t* removed);

/* vpp/algorithms/lbp/lbp_transform.hh:6-38 */
int orc_lbp_transform(const vpp_image_desc* out, const vpp_image_desc* in);
/* fast_detector/fast.hpp:511-551 (dense detector on the true ring, fast9_check_code :25-35) */
int orc_fast9_dense(const vpp_image_desc* out, const vpp_image_desc* in, int th);
/* fast_detector/fast.hpp:577-614 (in place) */
int orc_blockwise_maxima_filter(const vpp_image_desc* img, int block_size);
int orc_local_maxima_filter(const vpp_image_desc* img);

#ifdef __cplusplus
}
#endif
#endif
