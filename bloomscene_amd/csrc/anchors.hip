// Fused anchor expansion for gfx950 (include/bloomscene_anchors.h).
//
// The reference (gaussian_renderer/__init__.py:169-203) concatenates every per-candidate tensor
// into [N*K, 22], boolean-indexes it, splits it again and post-processes the pieces: ~10 torch
// kernels and ~25x the compulsory HBM traffic.  Here: one counting pass over neural_opacity, one
// tiny scan, and one kernel that reads each input once and writes each selected output once.
//
// Work partition (shared by all kernels, so the prefix sums agree): a workgroup of 256 lanes owns
// A = 256 / K whole anchors = A*K consecutive candidates; lane t < A*K is candidate base + t.
// Consecutive lanes touch consecutive rows of every per-candidate tensor (coalesced across the
// wave), and the K candidates of an anchor sit in one workgroup so the backward's per-anchor
// sums are an LDS reduction in slot order -- no atomics, deterministic.
#include "common.h"
#include "../../include/bloomscene_anchors.h"
#include "../../include/bloomscene_rast.h"

namespace bsr {

#define BSR_ANCHOR_BLOCK 256

// Exclusive prefix of `flag` over the workgroup's lanes (lane order), and the workgroup total.
__device__ __forceinline__ uint32_t wg_prefix(bool flag, uint32_t* s_wave, uint32_t& total)
{
	const uint64_t b = __ballot(flag);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t in_wave = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
	if (lane == 0) s_wave[wave] = (uint32_t)__popcll(b);
	__syncthreads();
	uint32_t before = 0, all = 0;
#pragma unroll
	for (int w = 0; w < BSR_ANCHOR_BLOCK / 64; w++) {
		const uint32_t c = s_wave[w];
		before += (w < wave) ? c : 0u;
		all += c;
	}
	total = all;
	return before + in_wave;
}

__global__ void __launch_bounds__(BSR_ANCHOR_BLOCK) k_anchor_select(int n_cand, int per_wg,
                                                                    const float* __restrict__ neural_opacity,
                                                                    uint8_t* __restrict__ mask,
                                                                    uint32_t* __restrict__ wg_count)
{
	__shared__ uint32_t s_wave[BSR_ANCHOR_BLOCK / 64];
	const int t = threadIdx.x;
	const long long i = (long long)blockIdx.x * per_wg + t;
	const bool live = t < per_wg && i < n_cand;
	const bool sel = live && neural_opacity[i] > 0.0f;
	if (live) mask[i] = sel ? 1 : 0;
	uint32_t total;
	wg_prefix(sel, s_wave, total);
	if (t == 0) wg_count[blockIdx.x] = total;
}

// In-place exclusive scan of wg_count[0..n) by one workgroup; the grand total goes to wg_count[n].
// Eight consecutive values per thread and round (two 16-B loads): 20 k workgroup counts take 3 rounds of
// barriers instead of 20 -- the kernel is pure latency.
__global__ void __launch_bounds__(1024) k_anchor_scan(int n, uint32_t* __restrict__ wg_count)
{
	__shared__ uint32_t s_wave[16];
	__shared__ uint32_t s_carry;
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	if (t == 0) s_carry = 0;
	__syncthreads();
	for (int base = 0; base < n; base += 8192) {
		const int i0 = base + t * 8;
		uint32_t v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) v[k] = (i0 + k < n) ? wg_count[i0 + k] : 0u;
		uint32_t mine = 0;
#pragma unroll
		for (int k = 0; k < 8; k++) mine += v[k];
		uint32_t incl = mine;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t o = __shfl_up(incl, d);
			if (lane >= d) incl += o;
		}
		if (lane == 63) s_wave[wave] = incl;
		__syncthreads();
		uint32_t before = s_carry, all = 0;
#pragma unroll
		for (int w = 0; w < 16; w++) {
			const uint32_t c = s_wave[w];
			before += (w < wave) ? c : 0u;
			all += c;
		}
		uint32_t run = before + incl - mine;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			if (i0 + k < n) wg_count[i0 + k] = run;
			run += v[k];
		}
		__syncthreads();
		if (t == 0) s_carry += all;
		__syncthreads();
	}
	if (t == 0) wg_count[n] = s_carry;
}

// ---- index list of the visible points (bsr_visible_filter_indices, SURVEY.md §8f rank 2, training half) ----
// radii > 0 per workgroup of 256 points -> counts; k_anchor_scan; ordered write of the indices.
__global__ void __launch_bounds__(BSR_ANCHOR_BLOCK) k_visible_count(int P, const int* __restrict__ radii,
                                                                    uint32_t* __restrict__ wg_count)
{
	__shared__ uint32_t s_wave[BSR_ANCHOR_BLOCK / 64];
	const long long i = (long long)blockIdx.x * BSR_ANCHOR_BLOCK + threadIdx.x;
	uint32_t total;
	wg_prefix(i < P && radii[i] > 0, s_wave, total);
	if (threadIdx.x == 0) wg_count[blockIdx.x] = total;
}

__global__ void __launch_bounds__(BSR_ANCHOR_BLOCK) k_visible_compact(int P, const int* __restrict__ radii,
                                                                      const uint32_t* __restrict__ wg_base,
                                                                      int* __restrict__ visible_idx)
{
	__shared__ uint32_t s_wave[BSR_ANCHOR_BLOCK / 64];
	const long long i = (long long)blockIdx.x * BSR_ANCHOR_BLOCK + threadIdx.x;
	const bool vis = i < P && radii[i] > 0;
	uint32_t total;
	const uint32_t rank = wg_prefix(vis, s_wave, total);
	if (vis) visible_idx[wg_base[blockIdx.x] + rank] = (int)i;
}

// torch.sigmoid in fp32: 1 / (1 + exp(-x))
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + bsr_expf(-x)); }

__global__ void __launch_bounds__(BSR_ANCHOR_BLOCK) k_anchor_expand(
    int n_cand, int K, int per_wg, const float* __restrict__ anchor, const float* __restrict__ grid_scaling,
    const float* __restrict__ grid_offsets, const float* __restrict__ neural_opacity, const float* __restrict__ color,
    const float* __restrict__ scale_rot, const uint32_t* __restrict__ wg_base, float* __restrict__ xyz,
    float* __restrict__ color_out, float* __restrict__ opacity, float* __restrict__ scaling, float* __restrict__ rot)
{
	__shared__ uint32_t s_wave[BSR_ANCHOR_BLOCK / 64];
	const int t = threadIdx.x;
	const long long i = (long long)blockIdx.x * per_wg + t;
	const bool live = t < per_wg && i < n_cand;
	const float nop = live ? neural_opacity[i] : 0.0f;
	const bool sel = live && nop > 0.0f;
	uint32_t total;
	const uint32_t rank = wg_prefix(sel, s_wave, total);
	if (!sel) return;
	const size_t dst = (size_t)wg_base[blockIdx.x] + rank;
	const long long n = i / K;
	const float* gs = grid_scaling + n * 6;
	const float* an = anchor + n * 3;
	const float* sr = scale_rot + i * 7;
	const float* of = grid_offsets + i * 3;
	const float* co = color + i * 3;
	opacity[dst] = nop;
	color_out[dst * 3 + 0] = co[0];
	color_out[dst * 3 + 1] = co[1];
	color_out[dst * 3 + 2] = co[2];
	scaling[dst * 3 + 0] = gs[3] * sigmoidf(sr[0]);
	scaling[dst * 3 + 1] = gs[4] * sigmoidf(sr[1]);
	scaling[dst * 3 + 2] = gs[5] * sigmoidf(sr[2]);
	const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
	const float nrm = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
	reinterpret_cast<float4*>(rot)[dst] = make_float4(q0 / nrm, q1 / nrm, q2 / nrm, q3 / nrm);
	xyz[dst * 3 + 0] = an[0] + of[0] * gs[0];
	xyz[dst * 3 + 1] = an[1] + of[1] * gs[1];
	xyz[dst * 3 + 2] = an[2] + of[2] * gs[2];
}

__global__ void __launch_bounds__(BSR_ANCHOR_BLOCK) k_anchor_expand_bwd(
    int n_cand, int n_anchors, int K, int per_wg, const float* __restrict__ grid_scaling,
    const float* __restrict__ grid_offsets, const float* __restrict__ neural_opacity,
    const float* __restrict__ scale_rot, const uint32_t* __restrict__ wg_base, const float* __restrict__ g_xyz,
    const float* __restrict__ g_color, const float* __restrict__ g_opacity, const float* __restrict__ g_scaling,
    const float* __restrict__ g_rot, float* __restrict__ d_anchor, float* __restrict__ d_gs,
    float* __restrict__ d_offsets, float* __restrict__ d_nop, float* __restrict__ d_color, float* __restrict__ d_sr)
{
	__shared__ uint32_t s_wave[BSR_ANCHOR_BLOCK / 64];
	__shared__ float s_part[BSR_ANCHOR_BLOCK][9];   // per candidate: d_anchor[3], d_gs[0:3], d_gs[3:6]
	const int t = threadIdx.x;
	const long long i = (long long)blockIdx.x * per_wg + t;
	const bool live = t < per_wg && i < n_cand;
	const bool sel = live && neural_opacity[i] > 0.0f;
	uint32_t total;
	const uint32_t rank = wg_prefix(sel, s_wave, total);
	float part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	if (live) {
		float dn = 0.0f, dc[3] = {0, 0, 0}, dsr[7] = {0, 0, 0, 0, 0, 0, 0}, dof[3] = {0, 0, 0};
		if (sel) {
			const size_t src = (size_t)wg_base[blockIdx.x] + rank;
			const long long n = i / K;
			const float* gs = grid_scaling + n * 6;
			const float* sr = scale_rot + i * 7;
			const float* of = grid_offsets + i * 3;
			if (g_opacity) dn = g_opacity[src];
			if (g_color) { dc[0] = g_color[src * 3]; dc[1] = g_color[src * 3 + 1]; dc[2] = g_color[src * 3 + 2]; }
			if (g_xyz) {
#pragma unroll
				for (int c = 0; c < 3; c++) {
					const float g = g_xyz[src * 3 + c];
					part[c] = g;               // d anchor
					part[3 + c] = g * of[c];   // d grid_scaling[0:3]
					dof[c] = g * gs[c];
				}
			}
			if (g_scaling) {
#pragma unroll
				for (int c = 0; c < 3; c++) {
					const float g = g_scaling[src * 3 + c];
					const float s = sigmoidf(sr[c]);
					part[6 + c] = g * s;                          // d grid_scaling[3:6]
					dsr[c] = g * gs[3 + c] * (s * (1.0f - s));
				}
			}
			if (g_rot) {
				// r = v / max(|v|, eps):  dv = (g - r (r.g)) / |v|  (|v| > eps),  g / eps otherwise
				const float4 g = reinterpret_cast<const float4*>(g_rot)[src];
				const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
				const float len = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
				if (len > 1e-12f) {
					const float inv = 1.0f / len;
					const float r0 = q0 * inv, r1 = q1 * inv, r2 = q2 * inv, r3 = q3 * inv;
					const float dot = r0 * g.x + r1 * g.y + r2 * g.z + r3 * g.w;
					dsr[3] = (g.x - r0 * dot) * inv;
					dsr[4] = (g.y - r1 * dot) * inv;
					dsr[5] = (g.z - r2 * dot) * inv;
					dsr[6] = (g.w - r3 * dot) * inv;
				} else {
					dsr[3] = g.x / 1e-12f; dsr[4] = g.y / 1e-12f; dsr[5] = g.z / 1e-12f; dsr[6] = g.w / 1e-12f;
				}
			}
		}
		d_nop[i] = dn;
#pragma unroll
		for (int c = 0; c < 3; c++) { d_color[i * 3 + c] = dc[c]; d_offsets[i * 3 + c] = dof[c]; }
#pragma unroll
		for (int c = 0; c < 7; c++) d_sr[i * 7 + c] = dsr[c];
	}
#pragma unroll
	for (int c = 0; c < 9; c++) s_part[t][c] = part[c];
	__syncthreads();
	// per-anchor sums over the K slots, slot order; lane t -> (anchor a = t / 9, component c = t % 9)
	const int A = per_wg / K;
	for (int u = t; u < A * 9; u += BSR_ANCHOR_BLOCK) {
		const int a = u / 9, c = u - a * 9;
		const long long n = (long long)blockIdx.x * A + a;
		if (n >= n_anchors) continue;
		float acc = 0.0f;
		for (int k = 0; k < K; k++) acc += s_part[a * K + k][c];
		if (c < 3) d_anchor[n * 3 + c] = acc;
		else d_gs[n * 6 + (c - 3)] = acc;
	}
}

}  // namespace bsr

using namespace bsr;

__global__ void __launch_bounds__(256) k_add_into(size_t n, const float* __restrict__ src, float* __restrict__ dst)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) dst[i] += src[i];
}

namespace {
struct Partition { int per_wg, n_wg; long long n_cand; };
inline bool make_partition(int N, int K, Partition* p)
{
	if (N < 0 || K <= 0 || K > BSR_ANCHOR_BLOCK) return false;
	const int A = BSR_ANCHOR_BLOCK / K;
	p->per_wg = A * K;
	p->n_wg = (N + A - 1) / A;
	p->n_cand = (long long)N * K;
	return p->n_cand <= 0x7fffffffLL;
}
}  // namespace

extern "C" {

size_t bsr_visible_scratch_bytes(int P)
{
	const size_t n_wg = ((size_t)(P > 0 ? P : 0) + BSR_ANCHOR_BLOCK - 1) / BSR_ANCHOR_BLOCK;
	return align_up((n_wg + 1) * sizeof(uint32_t), 256);
}

int bsr_visible_filter_indices(int P, int M, int width, int height, const float* means3D, const float* scales,
                               float scale_modifier, const float* rotations, const float* cov3D_precomp,
                               const float* viewmatrix, const float* projmatrix, float tan_fovx, float tan_fovy,
                               int prefiltered, int* radii, int* visible_idx, void* scratch, int* num_visible, int debug,
                               void* stream)
{
	if (!num_visible) return fail("bsr_visible_filter_indices: num_visible is NULL");
	*num_visible = 0;
	const int rc = bsr_visible_filter(P, M, width, height, means3D, scales, scale_modifier, rotations, cov3D_precomp,
	                                  viewmatrix, projmatrix, tan_fovx, tan_fovy, prefiltered, radii, debug, stream);
	if (rc != 0 || P == 0) return rc;
	if (!visible_idx || !scratch) return fail("bsr_visible_filter_indices: NULL buffer");
	hipStream_t s = (hipStream_t)stream;
	uint32_t* wg = (uint32_t*)scratch;
	const int n_wg = (P + BSR_ANCHOR_BLOCK - 1) / BSR_ANCHOR_BLOCK;
	hipLaunchKernelGGL(k_visible_count, dim3(n_wg), dim3(BSR_ANCHOR_BLOCK), 0, s, P, radii, wg);
	hipLaunchKernelGGL(k_anchor_scan, dim3(1), dim3(1024), 0, s, n_wg, wg);
	hipLaunchKernelGGL(k_visible_compact, dim3(n_wg), dim3(BSR_ANCHOR_BLOCK), 0, s, P, radii, wg, visible_idx);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return fail("bsr_visible_filter_indices: launch failed: %s", hipGetErrorString(e));
	uint32_t total = 0;
	if (read_u32_blocking(wg + n_wg, &total, s)) return 1;
	*num_visible = (int)total;
	return 0;
}

size_t bsr_anchor_scratch_bytes(int n_anchors, int n_offsets)
{
	Partition p;
	if (!make_partition(n_anchors, n_offsets, &p)) return 0;
	return align_up(((size_t)p.n_wg + 1) * sizeof(uint32_t), 256);
}

int bsr_anchor_select(int n_anchors, int n_offsets, const float* neural_opacity, uint8_t* mask, void* scratch,
                      int* num_selected, void* stream)
{
	Partition p;
	if (!make_partition(n_anchors, n_offsets, &p))
		return fail("bsr_anchor_select: need 0 < n_offsets <= %d and n_anchors * n_offsets < 2^31", BSR_ANCHOR_BLOCK);
	if (!num_selected) return fail("bsr_anchor_select: num_selected is NULL");
	*num_selected = 0;
	if (p.n_cand == 0) return 0;
	if (!neural_opacity || !mask || !scratch) return fail("bsr_anchor_select: NULL buffer");
	hipStream_t s = (hipStream_t)stream;
	uint32_t* wg = (uint32_t*)scratch;
	hipLaunchKernelGGL(k_anchor_select, dim3(p.n_wg), dim3(BSR_ANCHOR_BLOCK), 0, s, (int)p.n_cand, p.per_wg,
	                   neural_opacity, mask, wg);
	hipLaunchKernelGGL(k_anchor_scan, dim3(1), dim3(1024), 0, s, p.n_wg, wg);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return fail("bsr_anchor_select: launch failed: %s", hipGetErrorString(e));
	uint32_t total = 0;
	if (read_u32_blocking(wg + p.n_wg, &total, s)) return 1;
	*num_selected = (int)total;
	return 0;
}

int bsr_anchor_expand(int n_anchors, int n_offsets, int num_selected, const float* anchor, const float* grid_scaling,
                      const float* grid_offsets, const float* neural_opacity, const float* color,
                      const float* scale_rot, const void* scratch, float* xyz, float* color_out, float* opacity,
                      float* scaling, float* rot, void* stream)
{
	Partition p;
	if (!make_partition(n_anchors, n_offsets, &p)) return fail("bsr_anchor_expand: bad n_anchors / n_offsets");
	if (num_selected < 0 || num_selected > p.n_cand) return fail("bsr_anchor_expand: bad num_selected");
	if (p.n_cand == 0 || num_selected == 0) return 0;
	if (!anchor || !grid_scaling || !grid_offsets || !neural_opacity || !color || !scale_rot || !scratch || !xyz ||
	    !color_out || !opacity || !scaling || !rot)
		return fail("bsr_anchor_expand: NULL buffer");
	if (((uintptr_t)rot & 15) != 0) return fail("bsr_anchor_expand: rot must be 16-byte aligned");
	hipLaunchKernelGGL(k_anchor_expand, dim3(p.n_wg), dim3(BSR_ANCHOR_BLOCK), 0, (hipStream_t)stream, (int)p.n_cand,
	                   n_offsets, p.per_wg, anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot,
	                   (const uint32_t*)scratch, xyz, color_out, opacity, scaling, rot);
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) return fail("bsr_anchor_expand: launch failed: %s", hipGetErrorString(e));
	return 0;
}

int bsr_anchor_expand_backward(int n_anchors, int n_offsets, int num_selected, const float* grid_scaling,
                               const float* grid_offsets, const float* neural_opacity, const float* scale_rot,
                               const void* scratch, const float* dL_dxyz, const float* dL_dcolor_out,
                               const float* dL_dopacity, const float* dL_dscaling, const float* dL_drot,
                               float* dL_danchor, float* dL_dgrid_scaling, float* dL_dgrid_offsets,
                               float* dL_dneural_opacity, float* dL_dcolor, float* dL_dscale_rot, void* stream)
{
	Partition p;
	if (!make_partition(n_anchors, n_offsets, &p)) return fail("bsr_anchor_expand_backward: bad n_anchors / n_offsets");
	if (num_selected < 0 || num_selected > p.n_cand) return fail("bsr_anchor_expand_backward: bad num_selected");
	if (p.n_cand == 0) return 0;
	if (!grid_scaling || !grid_offsets || !neural_opacity || !scale_rot || !scratch || !dL_danchor ||
	    !dL_dgrid_scaling || !dL_dgrid_offsets || !dL_dneural_opacity || !dL_dcolor || !dL_dscale_rot)
		return fail("bsr_anchor_expand_backward: NULL buffer");
	if (dL_drot && ((uintptr_t)dL_drot & 15) != 0) return fail("bsr_anchor_expand_backward: dL_drot must be 16-byte aligned");
	hipLaunchKernelGGL(k_anchor_expand_bwd, dim3(p.n_wg), dim3(BSR_ANCHOR_BLOCK), 0, (hipStream_t)stream,
	                   (int)p.n_cand, n_anchors, n_offsets, p.per_wg, grid_scaling, grid_offsets, neural_opacity,
	                   scale_rot, (const uint32_t*)scratch, dL_dxyz, dL_dcolor_out, dL_dopacity, dL_dscaling, dL_drot,
	                   dL_danchor, dL_dgrid_scaling, dL_dgrid_offsets, dL_dneural_opacity, dL_dcolor, dL_dscale_rot);
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) return fail("bsr_anchor_expand_backward: launch failed: %s", hipGetErrorString(e));
	return 0;
}

// ---- the fused front end: selection + expansion + rasterizer in ONE native call ------------------------------------
// Layout (float offsets) of the packed per-Gaussian buffer of the S selected Gaussians: rot first, its rows are read
// as 16-byte vectors.  radii are int32.
//   rot[S,4] 0 | xyz[S,3] 4S | color[S,3] 7S | scaling[S,3] 10S | opacity[S] 13S | radii[S] 14S          (15 S words)
// and of the gradient scratch the backward fills:
//   dL_drot[S,4] 0 | dL_dxyz[S,3] 4S | dL_dcolor[S,3] 7S | dL_dscaling[S,3] 10S | dL_dopacity[S] 13S | dL_dmean2D[S,3] 14S   (17 S)
// BSR_FLAG_NO_READBACK: the packed buffer holds n_cand rows per section whatever the selection; rows [S, n_cand) (S on
// the device: the scan's grand total) become Gaussians no view can see -- centre AT the camera (view-space z = 0: behind
// the near plane, auxiliary.h:154), opacity 0 -- so the rasterizer culls them in its first test and their gradient rows
// are zeros.
__global__ void __launch_bounds__(256) k_anchor_pad(int n_cand, const uint32_t* __restrict__ total, const float* __restrict__ cam_pos,
                                                    float* __restrict__ xyz, float* __restrict__ rgb, float* __restrict__ opacity,
                                                    float* __restrict__ scaling, float* __restrict__ rot)
{
	const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_cand || i < (long long)*total) return;
	const float cx = cam_pos[0], cy = cam_pos[1], cz = cam_pos[2];
	xyz[3 * i] = cx; xyz[3 * i + 1] = cy; xyz[3 * i + 2] = cz;
	rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = 0.f;
	scaling[3 * i] = scaling[3 * i + 1] = scaling[3 * i + 2] = 0.f;
	opacity[i] = 0.f;
	rot[4 * i] = 1.f; rot[4 * i + 1] = rot[4 * i + 2] = rot[4 * i + 3] = 0.f;
}

size_t bsr_anchor_gaussian_bytes(int num_selected) { return (size_t)(num_selected > 0 ? num_selected : 0) * 15 * sizeof(float); }
size_t bsr_anchor_gradient_bytes(int num_selected) { return (size_t)(num_selected > 0 ? num_selected : 0) * 17 * sizeof(float); }

int bsr_anchor_render_forward(int n_anchors, int n_offsets, const float* anchor, const float* grid_scaling,
                              const float* grid_offsets, const float* neural_opacity, const float* color,
                              const float* scale_rot, uint8_t* mask, void* anchor_scratch, bsr_alloc_fn gaussianBuffer,
                              void* gaussian_user, bsr_alloc_fn geometryBuffer, void* geometry_user,
                              bsr_alloc_fn binningBuffer, void* binning_user, bsr_alloc_fn imageBuffer, void* image_user,
                              const float* background, int width, int height, float scale_modifier,
                              const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                              float tan_fovy, float* out_color, float* out_depth, int debug, void* stream,
                              int* num_selected, int* num_rendered, unsigned flags)
{
	if (!num_selected || !num_rendered) return fail("bsr_anchor_render_forward: num_selected / num_rendered is NULL");
	if (!gaussianBuffer) return fail("bsr_anchor_render_forward: gaussianBuffer callback is NULL");
	if (flags & BSR_FLAG_NO_READBACK) {
		// Static shapes, no host wait at all (include/bloomscene_anchors.h): every section of the packed buffer has
		// n_anchors * n_offsets rows, the selection stays on the device, the rows behind it are padded with invisible
		// Gaussians, and the rasterizer runs in its capacity mode on all of them.
		Partition p;
		if (!make_partition(n_anchors, n_offsets, &p))
			return fail("bsr_anchor_render_forward: need 0 < n_offsets <= %d and n_anchors * n_offsets < 2^31", BSR_ANCHOR_BLOCK);
		if (p.n_cand == 0) return fail("bsr_anchor_render_forward: BSR_FLAG_NO_READBACK needs at least one candidate");
		if (!neural_opacity || !mask || !anchor_scratch || !cam_pos) return fail("bsr_anchor_render_forward: NULL buffer");
		const int S_cap = (int)p.n_cand;
		float* g = (float*)gaussianBuffer(gaussian_user, bsr_anchor_gaussian_bytes(S_cap));
		if (!g) return fail("bsr_anchor_render_forward: gaussianBuffer returned null");
		if (((uintptr_t)g & 15) != 0) return fail("bsr_anchor_render_forward: gaussianBuffer must be 16-byte aligned");
		hipStream_t st = (hipStream_t)stream;
		uint32_t* wg = (uint32_t*)anchor_scratch;
		hipLaunchKernelGGL(k_anchor_select, dim3(p.n_wg), dim3(BSR_ANCHOR_BLOCK), 0, st, (int)p.n_cand, p.per_wg,
		                   neural_opacity, mask, wg);
		hipLaunchKernelGGL(k_anchor_scan, dim3(1), dim3(1024), 0, st, p.n_wg, wg);
		const size_t sc = (size_t)S_cap;
		float* rot = g, *xyz = g + 4 * sc, *rgb = g + 7 * sc, *scaling = g + 10 * sc, *opacity = g + 13 * sc;
		int* radii = (int*)(g + 14 * sc);
		if (bsr_anchor_expand(n_anchors, n_offsets, S_cap, anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot,
		                      anchor_scratch, xyz, rgb, opacity, scaling, rot, stream))
			return 1;
		hipLaunchKernelGGL(k_anchor_pad, dim3((unsigned)((sc + 255) / 256)), dim3(256), 0, st, S_cap, wg + p.n_wg, cam_pos, xyz, rgb,
		                   opacity, scaling, rot);
		if (hipGetLastError() != hipSuccess) return fail("bsr_anchor_render_forward: launch failed");
		*num_selected = S_cap;   // rows of every section = what the backward must be handed as num_selected
		return bsr_forward_ex(geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, S_cap, 1, 0,
		                      background, width, height, xyz, nullptr, rgb, opacity, scaling, scale_modifier, rot, nullptr,
		                      viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, 0, out_color, out_depth, radii, debug, stream,
		                      num_rendered, flags);
	}
	*num_rendered = 0;
	// While the GPU counts, ask for the S-sized buffer with a GUESS (the calling thread's previous selection of this
	// shape + 25 %, as the forward sizes its binning scratch): the callback -- interpreter time on the python host --
	// then runs beside the selection kernels instead of behind the blocking read.  A guess that turns out short costs
	// a second callback.  The layout inside the buffer depends on S alone.
	static thread_local int hint_N = -1, hint_K = -1, hint_S = 0;
	float* g = nullptr;
	long long cap = 0;
	if (hint_N == n_anchors && hint_K == n_offsets && hint_S > 0) {
		cap = (long long)hint_S + hint_S / 4 + 1024;
		if (cap > (long long)n_anchors * n_offsets) cap = (long long)n_anchors * n_offsets;
		g = (float*)gaussianBuffer(gaussian_user, bsr_anchor_gaussian_bytes((int)cap));
		if (!g) return fail("bsr_anchor_render_forward: gaussianBuffer returned null");
	}
	// the one blocking read of the selection (torch's boolean index has the same one); everything after it is
	// enqueued from here, without going back to the caller: expansion, preprocess, ... follow the count by microseconds
	if (bsr_anchor_select(n_anchors, n_offsets, neural_opacity, mask, anchor_scratch, num_selected, stream)) return 1;
	const int S = *num_selected;
	hint_N = n_anchors; hint_K = n_offsets; hint_S = S;
	if (S > 0 && (!g || S > cap)) {
		g = (float*)gaussianBuffer(gaussian_user, bsr_anchor_gaussian_bytes(S));
		if (!g) return fail("bsr_anchor_render_forward: gaussianBuffer returned null");
	}
	if (S > 0 && ((uintptr_t)g & 15) != 0) return fail("bsr_anchor_render_forward: gaussianBuffer must be 16-byte aligned");
	const size_t s = (size_t)S;
	float* rot = g, *xyz = g + 4 * s, *rgb = g + 7 * s, *scaling = g + 10 * s, *opacity = g + 13 * s;
	int* radii = (int*)(g + 14 * s);
	if (bsr_anchor_expand(n_anchors, n_offsets, S, anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot,
	                      anchor_scratch, xyz, rgb, opacity, scaling, rot, stream))
		return 1;
	// gaussian_renderer.render's call (GR:235-262): colors_precomp, sh_degree 1 (unused without SHs), prefiltered False
	return bsr_forward_ex(geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, S, 1, 0,
	                      background, width, height, S ? xyz : nullptr, nullptr, S ? rgb : nullptr, S ? opacity : nullptr,
	                      S ? scaling : nullptr, scale_modifier, S ? rot : nullptr, nullptr, viewmatrix, projmatrix,
	                      cam_pos, tan_fovx, tan_fovy, 0, out_color, out_depth, S ? radii : nullptr, debug, stream,
	                      num_rendered, flags);
}

int bsr_anchor_render_backward(int n_anchors, int n_offsets, int num_selected, int num_rendered,
                               const float* grid_scaling, const float* grid_offsets, const float* neural_opacity,
                               const float* scale_rot, const void* anchor_scratch, const float* gaussians,
                               char* geom_buffer, char* binning_buffer, char* image_buffer, const float* background,
                               int width, int height, float scale_modifier, const float* viewmatrix,
                               const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                               const float* dL_dpix, const float* out_depth, const float* dL_depths,
                               const float* g_xyz, const float* g_color, const float* g_opacity, const float* g_scaling,
                               const float* g_rot, float* gradient_scratch, float* dL_danchor, float* dL_dgrid_scaling,
                               float* dL_dgrid_offsets, float* dL_dneural_opacity, float* dL_dcolor,
                               float* dL_dscale_rot, int debug, void* stream, unsigned flags)
{
	const int S = num_selected;
	const size_t s = (size_t)(S > 0 ? S : 0);
	float* gs = gradient_scratch;
	if (S > 0) {
		if (!gaussians || !gs) return fail("bsr_anchor_render_backward: gaussians / gradient_scratch is NULL");
		if (((uintptr_t)gs & 15) != 0) return fail("bsr_anchor_render_backward: gradient_scratch must be 16-byte aligned");
		const float* rot = gaussians, *xyz = gaussians + 4 * s, *rgb = gaussians + 7 * s, *scaling = gaussians + 10 * s;
		const int* radii = (const int*)(gaussians + 14 * s);
		float* d_rot = gs, *d_xyz = gs + 4 * s, *d_rgb = gs + 7 * s, *d_scaling = gs + 10 * s, *d_opacity = gs + 13 * s,
		     *d_mean2D = gs + 14 * s;
		// (out_depth == NULL: the reference's backward; else the depth-gradient extension)
		const int rc = bsr_backward_ex(S, 1, 0, num_rendered, background, width, height, xyz, nullptr, rgb, scaling,
		                               scale_modifier, rot, nullptr, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
		                               radii, geom_buffer, binning_buffer, image_buffer, out_depth, dL_dpix, dL_depths,
		                               d_mean2D, nullptr, d_opacity, d_rgb, d_xyz, nullptr, nullptr, d_scaling, d_rot, debug,
		                               stream, flags);
		if (rc) return rc;
		// gradients that reached the expanded tensors from outside the rasterizer (BloomScene's scaling regulariser
		// reads `scaling`, bloomscene.py:296-297): added before they are pulled back through the expansion
		const float* extra[5] = {g_rot, g_xyz, g_color, g_scaling, g_opacity};
		float* into[5] = {d_rot, d_xyz, d_rgb, d_scaling, d_opacity};
		const size_t width_of[5] = {4, 3, 3, 3, 1};
		for (int i = 0; i < 5; i++)
			if (extra[i]) {
				const size_t n = s * width_of[i];
				hipLaunchKernelGGL(k_add_into, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,
				                   extra[i], into[i]);
			}
		if (hipGetLastError() != hipSuccess) return fail("bsr_anchor_render_backward: launch failed");
	}
	return bsr_anchor_expand_backward(n_anchors, n_offsets, S, grid_scaling, grid_offsets, neural_opacity, scale_rot,
	                                  anchor_scratch, S ? gs + 4 * s : nullptr, S ? gs + 7 * s : nullptr,
	                                  S ? gs + 13 * s : nullptr, S ? gs + 10 * s : nullptr, S ? gs : nullptr, dL_danchor,
	                                  dL_dgrid_scaling, dL_dgrid_offsets, dL_dneural_opacity, dL_dcolor, dL_dscale_rot,
	                                  stream);
}

}  // extern "C"
