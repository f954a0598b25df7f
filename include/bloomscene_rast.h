/*
 * bloomscene_rast.h -- C ABI of the MI355X-native (gfx950) differentiable Gaussian-splatting
 * rasterizer that drops in behind BloomScene's `depth_diff_gaussian_rasterization` module.
 *
 * Boundary: plain pointers, ints, floats and a hipStream_t passed as void*.  No torch types.
 * All pointers are DEVICE pointers unless stated otherwise; a NULL pointer means "optional input
 * absent" exactly like the empty CPU tensors -> nullptr convention of the reference
 * (depth_diff_gaussian_rasterization/__init__.py:198-208, rasterize_points.cu:95-113).
 * Every function returns 0 on success, non-zero on failure; bsr_last_error() then describes it.
 * Memory ownership is the reference's (rasterize_points.cu:27-33, rasterizer_impl.h:22-27): every byte of device
 * memory -- inputs, outputs, gradients and ALL scratch of forward and backward -- belongs to the caller; the library
 * allocates no device memory (no hipMalloc / hipMallocAsync on any path) and touches no allocator or memory-pool
 * setting.  The backward's 36 bytes (40 with the depth-gradient extension) per kept instance of partial sums live in
 * the binning buffer the forward sized, over sections that are dead by then.  The library keeps no results between
 * calls (per host thread: a pinned 32-byte HOST landing buffer -- which also receives the error word of a
 * prefiltered = 1 call to bsr_visible_filter straight from the kernel --, an event and the previous call's shape /
 * num_rendered -- and, for the fused anchor front end, its selection count -- as size hints; process-wide: the
 * opt-in stage profiler); the three scratch buffers handed from forward to backward
 * are opaque, as in the reference (__init__.py:97,106).
 * NUMERICS ARE PER CALL: the `flags` argument of bsr_forward_ex / bsr_backward_ex (BSR_FLAG_*).  There is no
 * process-wide numerics switch; two host threads may run different modes side by side (the reference interface has
 * no global mutable state either, rasterizer_impl.cu:228,241,286,435-437).
 *
 * Reference paths below are relative to
 *   /root/reference/submodules/depth-diff-gaussian-rasterization/
 */
#ifndef BLOOMSCENE_RAST_H_INCLUDED
#define BLOOMSCENE_RAST_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSR_VERSION 4

/* ---- per-call numerics flags (bsr_forward_ex, bsr_backward_ex, bsr_forward_views, the fused anchor front end) ------
 * BSR_FLAG_EXACT_EXP   forward: 0 (default) = the blend takes exp(power) from the hardware's v_exp_f32 (1 ulp) wherever
 *                      only its VALUE is needed, and from the library's pinned exp (the one the CPU oracle restates)
 *                      wherever the `alpha >= 1/255` decision of forward.cu:423-428 could depend on the last bit (a wave
 *                      with a pixel within 1.1e-3 of the cut in the exponent): every such decision, radii, num_rendered
 *                      and the per-tile lists are identical to the exact mode; colour / depth / final_T move by a few ulp
 *                      (the reference's own CUDA expf is a 2-ulp function), and the `T (1 - alpha) < 1e-4` stop of
 *                      :433-437 sees a T that differs by those ulps.  Set: the pinned exp on every evaluation -- the
 *                      forward then matches the CPU oracle bit for bit.  Accepted and ignored by the backward (a caller
 *                      may hand one flags word to both).
 * BSR_FLAG_EXACT_GRAD  backward: 0 (default) = k_render_bwd evaluates each (pixel, Gaussian) pair with hardware exp
 *                      outside the decision band, reciprocal + one refinement instead of the two divisions, fused
 *                      multiply-adds, moment sums and one projected accum_rec (DESIGN.md "Numerics").  Set: the reference's
 *                      per-pair operations on the reference's operands (backward.cu:521,527-536,557,561-583: IEEE
 *                      divisions, no contraction, the pinned exp, per-channel accum_rec, per-pair products) -- only the
 *                      ORDER of the sums then differs from a sequential evaluation; SURVEY.md 8(d)'s share of elements off
 *                      by more than 1e-4 falls below the summation-order floor.  Costs ~1.8x in that kernel.  Accepted
 *                      and ignored by the forward. */
#define BSR_FLAG_EXACT_EXP  1u
#define BSR_FLAG_EXACT_GRAD 2u
/* BSR_FLAG_NO_READBACK  forward (bsr_forward_ex): the call NEVER waits for the GPU.  The reference blocks on a 4-byte
 *   device-to-host copy of num_rendered to size its binning buffer (rasterizer_impl.cu:282), and so does the default
 *   path here (a 16-byte one, overlapped with the rest of the forward).  With this flag the CALLER states the size:
 *   on entry *num_rendered holds a CAPACITY (tile instances, > 0) -- the binning buffer is sized for it, every kernel
 *   takes the real count from device memory, and the call returns once everything is enqueued (nothing in it is illegal
 *   during hipStreamBeginCapture: a warmed-up forward + backward pair can be captured into a hipGraph and replayed --
 *   call bsr_check_deferred() AFTER the warm-up calls and BEFORE hipStreamBeginCapture: the warm-up leaves an overflow
 *   check pending, which is a blocking host wait; a forward that finds one pending while its stream is capturing
 *   returns an error instead of waiting).
 *   *num_rendered is left at the capacity: hand THAT to bsr_backward* as R (same scratch carve).
 *   Overflow (more instances kept than the capacity) is never silent: the frame's out_color / out_depth are filled with
 *   NaN by the tile kernel instead of being rendered, bsr_read_counts reports kept > capacity, and -- unless the stream
 *   was capturing -- the next bsr_forward* call of the same host thread (or bsr_check_deferred) returns an error naming
 *   both numbers.  A backward run on such a frame (handed the capacity as R) reads none of the missing lists: every
 *   Gaussian counts as culled and dL_dmean3D is filled with NaN.  The real counts of any forward, on demand (blocks):
 *   bsr_read_counts.
 *   Not for prefiltered calls (their violation flag is part of the read-back). */
#define BSR_FLAG_NO_READBACK 4u
/* TEST-ONLY flags of bsr_forward_ex (no reference counterpart).  They change NO result: they steer ONE call through code
 * the tests must reach and real inputs reach rarely -- per call, like every other flag (until round 6 these were
 * process-wide switches behind bsr_set_option; the library now keeps no switch between calls at all).
 *   BSR_FLAG_TEST_SORT_INT       every per-tile sort takes the integer compare-exchange flavour, which real inputs
 *                                reach only with NaN / non-positive depth bits (same order either way),
 *   BSR_FLAG_TEST_SORT_NETWORK   every per-tile sort of more than 64 keys runs the compare-exchange network; by default
 *                                a segment is first offered to the bucket-and-rank sort, which declines segments whose
 *                                depths pile up on one value (same order either way; implied by _SORT_INT),
 *   BSR_FLAG_TEST_SMALL_GRIDS    the grids of the wide per-tile sort classes are capped at 2 / 1 workgroups (product:
 *                                2560 / 512), so that a frame with a handful of long tiles exercises the loops in which
 *                                one workgroup sorts several tiles in turn (same order either way),
 *   BSR_FLAG_TEST_NO_HALF_MASKS  the forward keeps the per-half box tests of its tile walk to itself (by default it
 *                                leaves them in the top byte of the sorted id list -- one view of at most 2^24
 *                                Gaussians -- and the backward's waves build their lists from that byte instead of
 *                                testing every record again).
 * Accepted and ignored by the backward. */
#define BSR_FLAG_TEST_SORT_INT      0x100u
#define BSR_FLAG_TEST_SMALL_GRIDS   0x200u
#define BSR_FLAG_TEST_NO_HALF_MASKS 0x400u
#define BSR_FLAG_TEST_SORT_NETWORK  0x800u
#define BSR_FLAG_TEST_MASK          0xf00u

/* Resize callback for an opaque scratch buffer: must return a device pointer to at least
 * `bytes` bytes (256-byte aligned) that stays valid until the matching backward call.
 * Replaces std::function<char*(size_t)> geometryBuffer/binningBuffer/imageBuffer of
 * cuda_rasterizer/rasterizer.h:32-34 (built by resizeFunctional, rasterize_points.cu:27-33). */
typedef char* (*bsr_alloc_fn)(void* user, size_t bytes);

/* Library version (BSR_VERSION of the build). */
int bsr_version(void);

/* Message of the last failure on the calling thread ("" if none).  Replaces the exceptions
 * thrown by CHECK_CUDA (cuda_rasterizer/auxiliary.h:166-173) and AT_ERROR
 * (rasterize_points.cu:57-59). */
const char* bsr_last_error(void);

/* present[i] = (view-space z of means3D[i] > 0.2).  present is uint8[P] (torch.bool storage).
 * Replaces CudaRasterizer::Rasterizer::markVisible, cuda_rasterizer/rasterizer.h:24-29
 * (= rasterizer_impl.cu:141-153, kernel checkFrustum :54-66); bound by `_C.mark_visible`,
 * ext.cpp:19 / rasterize_points.cu:202-221. */
int bsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* Forward rasterization.  Writes out_color[3,H,W], out_depth[1,H,W], radii[P] (all fully
 * overwritten; radii may be NULL) and *num_rendered (HOST int: number of (Gaussian, tile)
 * instances).  Exactly one of shs / colors_precomp and exactly one of (scales, rotations) /
 * cov3D_precomp must be non-NULL.  background, viewmatrix, projmatrix, cam_pos are DEVICE
 * float[3]/[16]/[16]/[3].  debug != 0: synchronise and check after every stage.  prefiltered != 0:
 * a culled Gaussian is an error (the reference printf+__trap()s, auxiliary.h:156-160).
 * Performs one blocking 16-byte device->host read (the reference reads 4 bytes, rasterizer_impl.cu:282);
 * it is overlapped with the rest of the forward (binning, tile sort, render), for which binningBuffer may
 * be called with a size guessed from the calling thread's previous calls of the same (P, width,
 * height); if that cannot hold the instances actually kept, binningBuffer is called a second time
 * with the exact size and the tail of the pass is repeated.  (A guess that holds may be smaller than
 * bsr_binning_bytes(num_rendered): instances culled per tile need no room.) The last buffer returned is the one handed to backward.
 * *num_rendered is the reference's value (sum over Gaussians of the tiles of their bounding rect,
 * rasterizer_impl.cu:278-282); it sizes the binning scratch.  Instances whose tile the splat provably
 * cannot reach with alpha >= 1/255 are not listed internally (no output depends on them).
 * Replaces CudaRasterizer::Rasterizer::forward, cuda_rasterizer/rasterizer.h:31-54
 * (= rasterizer_impl.cu:198-339); bound by `_C.rasterize_gaussians`, ext.cpp:16 /
 * rasterize_points.cu:35-117. */
int bsr_forward(bsr_alloc_fn geometryBuffer, void* geometry_user,
                bsr_alloc_fn binningBuffer, void* binning_user,
                bsr_alloc_fn imageBuffer, void* image_user,
                int P, int D, int M,
                const float* background,
                int width, int height,
                const float* means3D,
                const float* shs,
                const float* colors_precomp,
                const float* opacities,
                const float* scales,
                float scale_modifier,
                const float* rotations,
                const float* cov3D_precomp,
                const float* viewmatrix,
                const float* projmatrix,
                const float* cam_pos,
                float tan_fovx, float tan_fovy,
                int prefiltered,
                float* out_color,
                float* out_depth,
                int* radii,
                int debug,
                void* stream,
                int* num_rendered);

/* bsr_forward with per-call numerics: flags = BSR_FLAG_EXACT_EXP or 0 (bsr_forward == flags 0).  Same reference
 * counterpart (cuda_rasterizer/rasterizer.h:31-54); the reference has one numerics mode, nvcc's expf. */
int bsr_forward_ex(bsr_alloc_fn geometryBuffer, void* geometry_user,
                   bsr_alloc_fn binningBuffer, void* binning_user,
                   bsr_alloc_fn imageBuffer, void* image_user,
                   int P, int D, int M,
                   const float* background,
                   int width, int height,
                   const float* means3D,
                   const float* shs,
                   const float* colors_precomp,
                   const float* opacities,
                   const float* scales,
                   float scale_modifier,
                   const float* rotations,
                   const float* cov3D_precomp,
                   const float* viewmatrix,
                   const float* projmatrix,
                   const float* cam_pos,
                   float tan_fovx, float tan_fovy,
                   int prefiltered,
                   float* out_color,
                   float* out_depth,
                   int* radii,
                   int debug,
                   void* stream,
                   int* num_rendered,
                   unsigned flags);

/* The counts of the forward call that filled `image_buffer` (for `width` x `height`), read back from it now: *kept =
 * tile instances after the exact tile cull (what the binning scratch must hold), *num_rendered = the reference's
 * num_rendered (sum of tile-rect areas).  Blocks until `stream` has reached the copy.  For BSR_FLAG_NO_READBACK callers:
 * kept > the capacity they passed means that frame was NOT rendered (NaN outputs); num_rendered sizes the next capacity.
 * No reference counterpart (the reference returns num_rendered from forward, rasterizer_impl.cu:282,339). */
int bsr_read_counts(const char* image_buffer, int width, int height, void* stream, int* kept, int* num_rendered);

/* Deferred status of the calling thread's last BSR_FLAG_NO_READBACK forward: 0 if it fit its capacity (or there was
 * none, or its stream was capturing), 1 + bsr_last_error() if it overflowed.  Waits for that forward's 16-byte copy
 * (long done once the frame has been consumed); clears the pending state.  Every bsr_forward* call makes this check
 * first. */
int bsr_check_deferred(void);

/* radii[P] of the Gaussians as the forward pass would compute them (0 = culled); nothing else.
 * Needs no scratch (the reference allocates and discards full state, rasterizer_impl.cu:361-375).
 * Replaces CudaRasterizer::Rasterizer::visible_filter, cuda_rasterizer/rasterizer.h:57-73
 * (= rasterizer_impl.cu:342-398); bound by `_C.rasterize_aussians_filter` [sic], ext.cpp:18 /
 * rasterize_points.cu:224-288. */
int bsr_visible_filter(int P, int M,
                       int width, int height,
                       const float* means3D,
                       const float* scales,
                       float scale_modifier,
                       const float* rotations,
                       const float* cov3D_precomp,
                       const float* viewmatrix,
                       const float* projmatrix,
                       float tan_fovx, float tan_fovy,
                       int prefiltered,
                       int* radii,
                       int debug,
                       void* stream);

/* bsr_forward for n_views cameras of one image size and field of view in ONE call (forward only: the
 * scratch it leaves is not a valid input of bsr_backward).  out_color[n_views,3,H,W], out_depth[n_views,1,H,W]
 * and radii[n_views,P] hold, view by view, exactly (bit for bit) what n_views calls of bsr_forward with
 * viewmatrices + 16*v, projmatrices + 16*v, cam_positions + 3*v (DEVICE float[n_views,16] / [n_views,16] /
 * [n_views,3]) write; *num_rendered is the sum of their num_rendered.  The views are stacked into one virtual
 * image of n_views * ceil(H/16) tile rows, so binning, per-tile sort and render run once over all of them:
 * the sparse views of a camera sweep (few visible Gaussians each) are launch/latency bound one at a time.
 * Scratch grows with n_views (geometry: n_views * P rows).  No reference counterpart: the reference renders
 * the rotate360 sweep view by view (bloomscene.py:191-193 -> gaussian_renderer/__init__.py:224-262).
 * flags: BSR_FLAG_EXACT_EXP or 0, as bsr_forward_ex. */
int bsr_forward_views(bsr_alloc_fn geometryBuffer, void* geometry_user,
                      bsr_alloc_fn binningBuffer, void* binning_user,
                      bsr_alloc_fn imageBuffer, void* image_user,
                      int P, int D, int M, int n_views,
                      const float* background,
                      int width, int height,
                      const float* means3D,
                      const float* shs,
                      const float* colors_precomp,
                      const float* opacities,
                      const float* scales,
                      float scale_modifier,
                      const float* rotations,
                      const float* cov3D_precomp,
                      const float* viewmatrices,
                      const float* projmatrices,
                      const float* cam_positions,
                      float tan_fovx, float tan_fovy,
                      int prefiltered,
                      float* out_color,
                      float* out_depth,
                      int* radii,
                      int debug,
                      void* stream,
                      int* num_rendered,
                      unsigned flags);

/* bsr_visible_filter plus the index list of the visible points: radii[P] exactly as bsr_visible_filter writes it,
 * visible_idx[P] (int32) whose first *num_visible (HOST int) entries are, ascending, the indices i with
 * radii[i] > 0 -- the rows a boolean index `x[radii > 0]` selects.  scratch: bsr_visible_scratch_bytes(P) bytes.
 * One blocking 4-byte device->host read.  No reference counterpart as ONE call: every training iteration the
 * reference runs visible_filter on the anchors (gaussian_renderer/__init__.py:342-349 from bloomscene.py:240) and then
 * boolean-indexes six per-anchor tensors with the mask (:33-43), each index with its own nonzero() pass and host
 * synchronisation; with the index list those become plain gathers (the per-iteration half). */
size_t bsr_visible_scratch_bytes(int P);
int bsr_visible_filter_indices(int P, int M,
                               int width, int height,
                               const float* means3D,
                               const float* scales,
                               float scale_modifier,
                               const float* rotations,
                               const float* cov3D_precomp,
                               const float* viewmatrix,
                               const float* projmatrix,
                               float tan_fovx, float tan_fovy,
                               int prefiltered,
                               int* radii,
                               int* visible_idx,
                               void* scratch,
                               int* num_visible,
                               int debug,
                               void* stream);

/* bsr_visible_filter for n_views cameras of one image size and field of view in ONE pass over the
 * Gaussians: radii[v*P + i] is exactly what bsr_visible_filter writes to radii[i] with
 * viewmatrices + 16*v / projmatrices + 16*v (DEVICE float[n_views,16] each).  Each Gaussian is read,
 * and its 3-D covariance built, once for all views.  No reference counterpart: the reference calls
 * visible_filter once per view of the rotate360 sweep (gaussian_renderer/__init__.py:342-347 from
 * bloomscene.py:191-193); SURVEY.md §8f rank 2. */
int bsr_visible_filter_views(int P, int n_views,
                             int width, int height,
                             const float* means3D,
                             const float* scales,
                             float scale_modifier,
                             const float* rotations,
                             const float* cov3D_precomp,
                             const float* viewmatrices,
                             const float* projmatrices,
                             float tan_fovx, float tan_fovy,
                             int* radii,
                             int debug,
                             void* stream);

/* EXTENSION (views.scatter_visible_gaussians): the same per-view test as bsr_visible_filter_views, reduced on the fly to
 * one mask per GROUP of views: group_mask[g][i] (uint8 [n_groups, P], fully written) = 1 iff some view v with
 * group_of_view[v] == g has radii > 0 for Gaussian i.  group_of_view: DEVICE int[n_views], values in [0, n_groups)
 * (only their low 6 bits are used),
 * n_groups <= 64.  With groups = the ranks of a view-parallel sweep this is "which Gaussians does rank g need": P
 * bytes per rank written instead of 4 P per view, and no radii > 0 / any() passes afterwards.
 * group_counts (DEVICE uint32[n_groups], may be NULL): the number of ones in each row of group_mask, written by this
 * call (what the caller needs to size its compaction: a 4*n_groups-byte read-back instead of a
 * reduction over the masks).  count_scratch: DEVICE, bsr_visible_groups_scratch_bytes(P, n_groups) bytes, needed
 * (and touched) only when group_counts is non-NULL -- per-workgroup partial counts; the caller's memory like every
 * other scratch of this library.  P == 0 or n_groups == 0: group_mask has no elements and is not touched;
 * group_counts (if given) is zeroed. */
size_t bsr_visible_groups_scratch_bytes(int P, int n_groups);
int bsr_visible_filter_groups(int P, int n_views, int n_groups,
                              int width, int height,
                              const float* means3D, const float* scales, float scale_modifier,
                              const float* rotations, const float* cov3D_precomp,
                              const float* viewmatrices, const float* projmatrices,
                              float tan_fovx, float tan_fovy,
                              const int* group_of_view, uint8_t* group_mask, uint32_t* group_counts,
                              void* count_scratch,
                              int debug, void* stream);

/* EXTENSION (views.scatter_visible_gaussians): dst[r] = the rows idx[r * idx_stride] of n_src (<= 8) per-Gaussian fp32
 * tensors laid side by side, dst row = sum(widths) floats (<= 4096): one pass instead of an index_select per tensor
 * plus a concatenation.  src / widths: HOST arrays (device pointers / floats per row of each tensor); idx: DEVICE int64
 * (idx_stride in elements: 2 reads the second column of a [R, 2] index-pair matrix in place); dst: DEVICE [R, sum(widths)].
 * P = rows of every source tensor: a row number outside [0, P) packs as zeros. */
int bsr_pack_rows(int R, int P, int n_src,
                  const float* const* src, const int* widths,
                  const int64_t* idx, int idx_stride,
                  float* dst,
                  int debug, void* stream);

/* EXTENSION (views.render_views_sharded(compact=True)): the same gather with one destination per tensor:
 * dst[k] (HOST array of DEVICE pointers) = [R, widths[k]] = rows idx[] of src[k]. */
int bsr_gather_rows(int R, int P, int n_src,
                    const float* const* src, const int* widths,
                    const int64_t* idx, int idx_stride,
                    float* const* dst,
                    int debug, void* stream);

/* Backward pass for the forward call that produced (radii, geom/binning/image buffers, R).
 * dL_dpix is [3,H,W]; dL_depths [1,H,W] is accepted and ignored exactly like the reference
 * (backward.cu:457-463,539-554).  All nine gradient outputs are FULLY OVERWRITTEN (no pre-zeroing
 * needed): dL_dmean2D[P,3], dL_dconic[P,2,2] (internal, but part of the reference signature),
 * dL_dopacity[P,1], dL_dcolor[P,3], dL_dmean3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3],
 * dL_dscale[P,3], dL_drot[P,4]; rows of culled Gaussians are zero.  dL_dsh may be NULL when
 * M == 0; dL_dscale/dL_drot are written only when scales != NULL.  dL_dconic may be NULL, and so may
 * dL_dcolor when shs != NULL and dL_dcov3D when scales != NULL: they are then intermediate results the
 * reference materialises (rasterize_points.cu:154-162) but nobody reads, and are simply not written.
 * The call WRITES into the forward's saved buffers (the slab in the binning buffer, the tail-pool counter in the image
 * buffer): two backward calls on the SAME forward state must be stream-ordered (one after the other on one stream, or
 * joined by an event) -- run concurrently on two streams they would share the counter and both be wrong.  Sequential
 * repeats (retain_graph) are fine.
 * Replaces CudaRasterizer::Rasterizer::backward, cuda_rasterizer/rasterizer.h:75-105
 * (= rasterizer_impl.cu:403-504); bound by `_C.rasterize_gaussians_backward`, ext.cpp:17 /
 * rasterize_points.cu:119-200. */
int bsr_backward(int P, int D, int M, int R,
                 const float* background,
                 int width, int height,
                 const float* means3D,
                 const float* shs,
                 const float* colors_precomp,
                 const float* scales,
                 float scale_modifier,
                 const float* rotations,
                 const float* cov3D_precomp,
                 const float* viewmatrix,
                 const float* projmatrix,
                 const float* campos,
                 float tan_fovx, float tan_fovy,
                 const int* radii,
                 char* geom_buffer,
                 char* binning_buffer,
                 char* image_buffer,
                 const float* dL_dpix,
                 const float* dL_depths,
                 float* dL_dmean2D,
                 float* dL_dconic,
                 float* dL_dopacity,
                 float* dL_dcolor,
                 float* dL_dmean3D,
                 float* dL_dcov3D,
                 float* dL_dsh,
                 float* dL_dscale,
                 float* dL_drot,
                 int debug,
                 void* stream);

/* EXTENSION (no reference counterpart): bsr_backward that ALSO differentiates
 * the depth target.  The reference forward normalises out_depth = D / acc where acc > 0.5 (else 0;
 * forward.cu:459-468) but its backward drops dL_depths (backward.cu:457-463,539-554), so BloomScene's
 * depth regularisers (bloomscene.py:307-325) never reach the Gaussians.  This entry point adds the
 * true derivative of that forward -- through alpha into mean2D / conic / opacity and through the
 * view-space z into dL_dmean3D -- to everything bsr_backward computes.  out_depth is the [1,H,W]
 * depth image the matching bsr_forward wrote.  Same outputs and overwrite rules as bsr_backward. */
int bsr_backward_depth(int P, int D, int M, int R,
                       const float* background,
                       int width, int height,
                       const float* means3D,
                       const float* shs,
                       const float* colors_precomp,
                       const float* scales,
                       float scale_modifier,
                       const float* rotations,
                       const float* cov3D_precomp,
                       const float* viewmatrix,
                       const float* projmatrix,
                       const float* campos,
                       float tan_fovx, float tan_fovy,
                       const int* radii,
                       char* geom_buffer,
                       char* binning_buffer,
                       char* image_buffer,
                       const float* out_depth,
                       const float* dL_dpix,
                       const float* dL_depths,
                       float* dL_dmean2D,
                       float* dL_dconic,
                       float* dL_dopacity,
                       float* dL_dcolor,
                       float* dL_dmean3D,
                       float* dL_dcov3D,
                       float* dL_dsh,
                       float* dL_dscale,
                       float* dL_drot,
                       int debug,
                       void* stream);

/* bsr_backward / bsr_backward_depth with per-call numerics.  out_depth == NULL: the reference's backward (dL_depths
 * accepted and ignored, == bsr_backward); non-NULL: the depth-gradient extension (== bsr_backward_depth).
 * flags: BSR_FLAG_EXACT_GRAD or 0 (BSR_FLAG_EXACT_EXP is accepted and ignored).  Same reference counterpart as
 * bsr_backward (cuda_rasterizer/rasterizer.h:75-105; the per-pair operations BSR_FLAG_EXACT_GRAD restores are
 * backward.cu:521,527-536,557,561-583). */
int bsr_backward_ex(int P, int D, int M, int R,
                    const float* background,
                    int width, int height,
                    const float* means3D,
                    const float* shs,
                    const float* colors_precomp,
                    const float* scales,
                    float scale_modifier,
                    const float* rotations,
                    const float* cov3D_precomp,
                    const float* viewmatrix,
                    const float* projmatrix,
                    const float* campos,
                    float tan_fovx, float tan_fovy,
                    const int* radii,
                    char* geom_buffer,
                    char* binning_buffer,
                    char* image_buffer,
                    const float* out_depth,
                    const float* dL_dpix,
                    const float* dL_depths,
                    float* dL_dmean2D,
                    float* dL_dconic,
                    float* dL_dopacity,
                    float* dL_dcolor,
                    float* dL_dmean3D,
                    float* dL_dcov3D,
                    float* dL_dsh,
                    float* dL_dscale,
                    float* dL_drot,
                    int debug,
                    void* stream,
                    unsigned flags);

/* Scratch sizes, for callers that pre-allocate instead of growing inside the callback
 * (the reference's required<T>(n), cuda_rasterizer/rasterizer_impl.h:68-73). */
size_t bsr_geometry_bytes(int P);
size_t bsr_binning_bytes(int num_rendered);   /* 44 B / instance + 2 MB: point list (4 B) + radix ping-pong buffers (2 x 12 B) or, over the same bytes, the backward slab (40 B) */
size_t bsr_image_bytes(int width, int height);

/* Byte offset, inside the image buffer a forward call filled, of its float final_T[height * width] (the
 * transmittance left at each pixel; alpha = 1 - final_T).  No reference counterpart: the reference keeps
 * accum_alpha in its opaque ImageState (rasterizer_impl.h:45-53, forward.cu:459) and exposes no alpha output;
 * used by the host's opt-in return_alpha extension. */
size_t bsr_transmittance_offset(const void* image_buffer);

/* ---- measurement hooks (no reference counterpart; used by bench.py only) ------------------
 * bsr_profile_enable(1) makes every kernel stage of subsequent calls be bracketed by hipEvents
 * on the call's stream (bsr_profile_enable(N), N > 1: only of every Nth forward call and
 * of the backward call that follows it -- each event costs a few microseconds of pipeline bubble); bsr_profile_read() synchronises those events and returns, per stage
 * name, the accumulated milliseconds and launch count since the last bsr_profile_reset(). */
#define BSR_PROFILE_MAX_STAGES 16
typedef struct bsr_stage_profile {
	const char* name;
	double total_ms;
	int launches;
} bsr_stage_profile;
int bsr_profile_enable(int on);
/* Restrict the bracketing to ONE stage ("render_bwd", "render_fwd", "preprocess", ...; NULL or "" = all stages
 * again): two events per sampled call instead of fourteen, for timing the dominant kernel inside a throughput
 * measurement without taxing it (an event pair costs ~35 us of pipeline bubble on MI355X). */
int bsr_profile_only(const char* stage);
int bsr_profile_reset(void);
int bsr_profile_read(bsr_stage_profile* out, int max_stages);

#ifdef __cplusplus
}
#endif
#endif
