/*
 * bloomscene_anchors.h -- C ABI of the fused "anchor expansion" that feeds the rasterizer
 * (SURVEY.md §8f rank 1): the tail of BloomScene's generate_neural_gaussians,
 *   /root/reference/gaussian_renderer/__init__.py:169-203
 * which the reference runs as ~10 small torch kernels around a [N*K, 22] concat + boolean index.
 *
 * Per anchor n (N of them) and offset slot k (K = n_offsets) there is one CANDIDATE Gaussian
 * i = n*K + k.  A candidate is SELECTED when neural_opacity[i] > 0 (GR:169-171; the caller has
 * already multiplied in the binary grid mask, GR:168).  Selected candidates are written densely, in
 * candidate order (what torch's boolean indexing produces, GR:192):
 *
 *   opacity  = neural_opacity[i]                                    (GR:174)
 *   color    = color[i]                                             (GR:177-178,193)
 *   scaling  = grid_scaling[n, 3:6] * sigmoid(scale_rot[i, 0:3])    (GR:196-197)
 *   rot      = scale_rot[i, 3:7] / max(||scale_rot[i, 3:7]||, 1e-12)  (GR:198; torch F.normalize)
 *   xyz      = anchor[n] + grid_offsets[n, k] * grid_scaling[n, 0:3]  (GR:200-201)
 *
 * Boundary rules are those of bloomscene_rast.h: plain DEVICE pointers and ints, a hipStream_t
 * passed as void*, 0 on success, bsr_last_error() on failure, no state kept between calls.
 * All tensors are dense row-major fp32: anchor[N,3], grid_scaling[N,6], grid_offsets[N,K,3],
 * neural_opacity[N*K], color[N*K,3], scale_rot[N*K,7].
 */
#ifndef BLOOMSCENE_ANCHORS_H_INCLUDED
#define BLOOMSCENE_ANCHORS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#include "bloomscene_rast.h"   /* bsr_alloc_fn */

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of the scratch buffer that carries the selection's prefix sums from bsr_anchor_select to
 * bsr_anchor_expand / bsr_anchor_expand_backward (opaque; 0 < n_offsets <= 256). */
size_t bsr_anchor_scratch_bytes(int n_anchors, int n_offsets);

/* mask[i] = neural_opacity[i] > 0 for the N*K candidates (uint8, torch.bool storage) and
 * *num_selected (HOST int) = how many are set.  Fills `scratch`.  One blocking 4-byte
 * device->host read -- the same synchronisation torch's `neural_opacity[mask]` performs.
 * Replaces GR:169-171 (`mask = (neural_opacity > 0.0).view(-1)`) and the nonzero() pass hidden in
 * every boolean index of GR:174,192. */
int bsr_anchor_select(int n_anchors, int n_offsets, const float* neural_opacity, uint8_t* mask, void* scratch,
                      int* num_selected, void* stream);

/* Writes the num_selected selected Gaussians: xyz[S,3], color_out[S,3], opacity[S,1], scaling[S,3],
 * rot[S,4] (S = num_selected as returned by bsr_anchor_select on the same neural_opacity/scratch).
 * Replaces GR:174 and GR:183-201. */
int bsr_anchor_expand(int n_anchors, int n_offsets, int num_selected,
                      const float* anchor, const float* grid_scaling, const float* grid_offsets,
                      const float* neural_opacity, const float* color, const float* scale_rot,
                      const void* scratch,
                      float* xyz, float* color_out, float* opacity, float* scaling, float* rot,
                      void* stream);

/* Gradient of bsr_anchor_expand: what torch.autograd derives for GR:174-201.  Any of the five
 * upstream gradients may be NULL (= zeros).  All six outputs are FULLY OVERWRITTEN:
 * dL_danchor[N,3], dL_dgrid_scaling[N,6], dL_dgrid_offsets[N,K,3], dL_dneural_opacity[N*K],
 * dL_dcolor[N*K,3], dL_dscale_rot[N*K,7]; rows of unselected candidates are zero.  The sums over
 * the K offsets of an anchor are formed in slot order (deterministic; torch's index_put_ /
 * repeat backward uses atomics). */
int bsr_anchor_expand_backward(int n_anchors, int n_offsets, int num_selected,
                               const float* grid_scaling, const float* grid_offsets,
                               const float* neural_opacity, const float* scale_rot,
                               const void* scratch,
                               const float* dL_dxyz, const float* dL_dcolor_out, const float* dL_dopacity,
                               const float* dL_dscaling, const float* dL_drot,
                               float* dL_danchor, float* dL_dgrid_scaling, float* dL_dgrid_offsets,
                               float* dL_dneural_opacity, float* dL_dcolor, float* dL_dscale_rot,
                               void* stream);

/* ---- the fused front end: selection + expansion + rasterizer forward (and their backward) in ONE native call --------
 * gaussian_renderer.render on the anchor representation (GR:165-203 then GR:235-262) needs the number S of selected
 * Gaussians on the HOST (it shapes every tensor downstream), and the host is blocked on that read while the GPU waits
 * for whatever is launched next.  Called separately (bsr_anchor_select, bsr_anchor_expand, bsr_forward), the
 * interpreter's work between the three calls is GPU idle time: 100 us of a 630 us step at 100 k anchors x 10, 512^2.
 * Here the library reads S, asks for the one S-sized buffer through `gaussianBuffer`, and enqueues the expansion and
 * the whole rasterizer forward without returning to the caller.
 *
 * gaussianBuffer may be called BEFORE S is known, with a size guessed from the calling thread's previous call of the
 * same (N, K), and a second time with bsr_anchor_gaussian_bytes(S) if that guess was short (the callback's time then
 * overlaps the selection kernels); the last buffer returned is the one used, and it holds >= bsr_anchor_gaussian_bytes(S).
 * gaussianBuffer(user, bytes) must return 16-byte aligned device memory; the library lays the
 * selected Gaussians out in it as fp32 words  rot[S,4] at 0 | xyz[S,3] at 4S | color[S,3] at 7S | scaling[S,3] at 10S |
 * opacity[S] at 13S | radii[S] (int32) at 14S,  i.e. the outputs of bsr_anchor_expand and the radii of bsr_forward.
 * The rasterizer call is the reference's: colors_precomp = color, sh_degree 1, prefiltered False (GR:244-262).
 * Everything else as bsr_anchor_select / bsr_anchor_expand / bsr_forward.
 *
 * flags & BSR_FLAG_NO_READBACK (include/bloomscene_rast.h): STATIC SHAPES, no host wait anywhere -- neither for S nor for
 * num_rendered.  gaussianBuffer is asked once for bsr_anchor_gaussian_bytes(N * K); every section has N * K rows
 * (layout above with S = N * K); the S selected Gaussians fill the first rows of each section in the usual order and the
 * rows behind them are padded with Gaussians no camera can see (centre at cam_pos, opacity 0: culled by the rasterizer's
 * near-plane test, radii 0, zero gradient rows).  S itself stays on the device (word n_wg of anchor_scratch; the mask
 * holds the selection).  On entry *num_rendered = the rasterizer's capacity (tile instances); on return *num_selected
 * = N * K and *num_rendered = that capacity: hand both to bsr_anchor_render_backward.  The frame and every gradient of
 * the selected rows are bit-identical to the default path's.  Capturable into a hipGraph (after one warm-up call followed
 * by bsr_check_deferred(): see BSR_FLAG_NO_READBACK in bloomscene_rast.h). */
size_t bsr_anchor_gaussian_bytes(int num_selected);
int bsr_anchor_render_forward(int n_anchors, int n_offsets,
                              const float* anchor, const float* grid_scaling, const float* grid_offsets,
                              const float* neural_opacity, const float* color, const float* scale_rot,
                              uint8_t* mask, void* anchor_scratch,
                              bsr_alloc_fn gaussianBuffer, void* gaussian_user,
                              bsr_alloc_fn geometryBuffer, void* geometry_user,
                              bsr_alloc_fn binningBuffer, void* binning_user,
                              bsr_alloc_fn imageBuffer, void* image_user,
                              const float* background, int width, int height, float scale_modifier,
                              const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                              float tan_fovx, float tan_fovy,
                              float* out_color, float* out_depth, int debug, void* stream,
                              int* num_selected, int* num_rendered, unsigned flags);

/* Backward of bsr_anchor_render_forward: bsr_backward (bsr_backward_depth when out_depth is non-NULL) into
 * `gradient_scratch`, then bsr_anchor_expand_backward from it.  gradient_scratch: bsr_anchor_gradient_bytes(S) bytes,
 * 16-byte aligned, laid out as  dL_drot[S,4] at 0 | dL_dxyz[S,3] at 4S | dL_dcolor[S,3] at 7S | dL_dscaling[S,3] at 10S |
 * dL_dopacity[S] at 13S | dL_dmean2D[S,3] at 14S;  it is left filled -- dL_dmean2D is the screen-space gradient the
 * caller reads through `viewspace_points.grad` (GR:224-229, scene/gaussian_model.py:756).
 * g_xyz .. g_rot (each may be NULL): gradients that reached the expanded tensors from OUTSIDE the rasterizer, added to
 * the rasterizer's before the expansion is differentiated (BloomScene's scaling regulariser reads `scaling`). */
size_t bsr_anchor_gradient_bytes(int num_selected);
int bsr_anchor_render_backward(int n_anchors, int n_offsets, int num_selected, int num_rendered,
                               const float* grid_scaling, const float* grid_offsets,
                               const float* neural_opacity, const float* scale_rot,
                               const void* anchor_scratch, const float* gaussians,
                               char* geom_buffer, char* binning_buffer, char* image_buffer,
                               const float* background, int width, int height, float scale_modifier,
                               const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                               float tan_fovx, float tan_fovy,
                               const float* dL_dpix, const float* out_depth, const float* dL_depths,
                               const float* g_xyz, const float* g_color, const float* g_opacity,
                               const float* g_scaling, const float* g_rot,
                               float* gradient_scratch,
                               float* dL_danchor, float* dL_dgrid_scaling, float* dL_dgrid_offsets,
                               float* dL_dneural_opacity, float* dL_dcolor, float* dL_dscale_rot,
                               int debug, void* stream, unsigned flags);

#ifdef __cplusplus
}
#endif
#endif
