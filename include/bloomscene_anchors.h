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

#ifdef __cplusplus
}
#endif
#endif
