// Per-Gaussian front end: view-space projection, near cull, cov3D/cov2D/conic, radius, tile rect,
// SH -> RGB, per-tile instance counting.  One thread per Gaussian (HBM-bound streaming kernel).
//
// Semantics follow the reference kernels preprocessCUDA / filter_preprocessCUDA / checkFrustum
// (cuda_rasterizer/forward.cu:155-335, rasterizer_impl.cu:54-66 under
// /root/reference/submodules/depth-diff-gaussian-rasterization); the GLM column-major products
// are written out with their zero terms dropped, keeping the order of the remaining operations.
#include "common.h"
#include "sh.h"
#include "tile_common.h"

namespace bsr {

struct Cov2DTerms {
	float t0[3], t1[3];   // T[0][k], T[1][k] in the reference's glm indexing (rows of J*Rwc)
	float tx, ty, tz;     // clamped view-space mean
	float xmul, ymul;     // 0 where the +-1.3 guard-band clamp was active
};

// reference forward.cu:74-113 / backward.cu:162-194 (shared prologue)
__device__ __forceinline__ void cov2d_terms(const float3 mean, float focal_x, float focal_y, float tan_fovx,
                                            float tan_fovy, const float* __restrict__ vm, Cov2DTerms& o)
{
	float tx = vm[0] * mean.x + vm[4] * mean.y + vm[8] * mean.z + vm[12];
	float ty = vm[1] * mean.x + vm[5] * mean.y + vm[9] * mean.z + vm[13];
	const float tz = vm[2] * mean.x + vm[6] * mean.y + vm[10] * mean.z + vm[14];
	const float limx = 1.3f * tan_fovx;
	const float limy = 1.3f * tan_fovy;
	const float txtz = tx / tz;
	const float tytz = ty / tz;
	tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
	ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
	o.xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
	o.ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
	const float J00 = focal_x / tz;
	const float J02 = -(focal_x * tx) / (tz * tz);
	const float J11 = focal_y / tz;
	const float J12 = -(focal_y * ty) / (tz * tz);
	// T = W * J with W[0]=(vm0,vm4,vm8), W[1]=(vm1,vm5,vm9), W[2]=(vm2,vm6,vm10)
	o.t0[0] = vm[0] * J00 + vm[2] * J02;
	o.t0[1] = vm[4] * J00 + vm[6] * J02;
	o.t0[2] = vm[8] * J00 + vm[10] * J02;
	o.t1[0] = vm[1] * J11 + vm[2] * J12;
	o.t1[1] = vm[5] * J11 + vm[6] * J12;
	o.t1[2] = vm[9] * J11 + vm[10] * J12;
	o.tx = tx; o.ty = ty; o.tz = tz;
}

// cov = T^t * Vrk^t * T, entries (0,0), (0,1), (1,1), +0.3 low-pass (forward.cu:104-112)
__device__ __forceinline__ void cov2d_eval(const Cov2DTerms& t, const float* c, float& a, float& b, float& cc)
{
	// A[col][0] = T0 . V[:,col],  A[col][1] = T1 . V[:,col]
	const float A00 = t.t0[0] * c[0] + t.t0[1] * c[1] + t.t0[2] * c[2];
	const float A10 = t.t0[0] * c[1] + t.t0[1] * c[3] + t.t0[2] * c[4];
	const float A20 = t.t0[0] * c[2] + t.t0[1] * c[4] + t.t0[2] * c[5];
	const float A01 = t.t1[0] * c[0] + t.t1[1] * c[1] + t.t1[2] * c[2];
	const float A11 = t.t1[0] * c[1] + t.t1[1] * c[3] + t.t1[2] * c[4];
	const float A21 = t.t1[0] * c[2] + t.t1[1] * c[4] + t.t1[2] * c[5];
	a = A00 * t.t0[0] + A10 * t.t0[1] + A20 * t.t0[2];
	b = A01 * t.t0[0] + A11 * t.t0[1] + A21 * t.t0[2];
	cc = A01 * t.t1[0] + A11 * t.t1[1] + A21 * t.t1[2];
	a += 0.3f;
	cc += 0.3f;
}

// reference auxiliary.h:41-44 (double-precision literals)
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// reference auxiliary.h:46-56
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, int* rmin, int* rmax)
{
	rmin[0] = min(gx, max(0, (int)((px - max_radius) / BSR_TILE)));
	rmin[1] = min(gy, max(0, (int)((py - max_radius) / BSR_TILE)));
	rmax[0] = min(gx, max(0, (int)((px + max_radius + BSR_TILE - 1) / BSR_TILE)));
	rmax[1] = min(gy, max(0, (int)((py + max_radius + BSR_TILE - 1) / BSR_TILE)));
}


// COLOR: 0 = any SH shape or precomputed colours; 1 = only SH of active degree 3 with M = 16 (streamed evaluation,
// sh.h); 2 = only precomputed colours (BloomScene's call shape).  The specialised instantiations exist for their
// register count: the generic kernel carries the 48-coefficient path (92 VGPRs, 5 waves per SIMD), they run at 8.
template <bool FILTER_ONLY, int COLOR = 0>
__global__ void __launch_bounds__(256) k_preprocess(const PreArgs a)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	const bool in_range = idx < a.P;
	if (FILTER_ONLY && !in_range) return;   // the full kernel keeps every thread for the workgroup scan below
	// view-batched call: grid.y = view; `id` indexes the geometry rows (== idx for a single view)
	const int view = (int)blockIdx.y;
	const int wg = view * (int)gridDim.x + (int)blockIdx.x;
	const int id = wg * 256 + (int)threadIdx.x;
	const int ty_off = view * a.gy;   // tile rows of this view in the stacked virtual image

	// histogram of the kept instances over the low 8 bits of their tile id = the per-workgroup histogram of
	// the first radix pass of the binning (k_emit_scatter), counted while the tiles are being tested anyway
	__shared__ uint32_t s_hist[256];
	if (!FILTER_ONLY) {
		s_hist[threadIdx.x] = 0;
		__syncthreads();
	}

	int radius_out = 0;
	ushort4 rect_out = make_ushort4(0, 0, 0, 0);
	uint64_t kept_mask = 0;
	float4 rq0 = make_float4(0.f, 0.f, 0.f, 0.f), rq1 = rq0, rq2 = rq0;   // splat record, stored at the end

	const int ld = in_range ? idx : 0;
	float3 p = make_float3(a.means3D[3 * ld], a.means3D[3 * ld + 1], a.means3D[3 * ld + 2]);
	const float* vm = a.viewmatrix + 16 * view;
	const float* pm = a.projmatrix + 16 * view;
	float pvz = vm[2] * p.x + vm[6] * p.y + vm[10] * p.z + vm[14];
	const bool alive = in_range && !(pvz <= BSR_NEAR);   // reference auxiliary.h:154
	if (in_range && !alive && a.prefiltered) a.flags[0] = 1;

	// ---- part 1, per Gaussian: everything up to the tile rect (reference forward.cu:186-236).  `vis` = the rect is
	// not empty; what part 2 needs of a visible Gaussian stays in these variables.
	bool vis = false;
	float cov3D[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	float conic_a = 0.f, conic_b = 0.f, conic_c = 0.f, pix_x = 0.f, pix_y = 0.f;
	int rmin[2] = {0, 0}, rmax[2] = {0, 0};
	if (alive) {
		const float hx = pm[0] * p.x + pm[4] * p.y + pm[8] * p.z + pm[12];
		const float hy = pm[1] * p.x + pm[5] * p.y + pm[9] * p.z + pm[13];
		const float hw = pm[3] * p.x + pm[7] * p.y + pm[11] * p.z + pm[15];
		const float p_w = 1.0f / (hw + 0.0000001f);
		const float projx = hx * p_w, projy = hy * p_w;

		if (a.cov3D_precomp != nullptr) {
#pragma unroll
			for (int k = 0; k < 6; k++) cov3D[k] = a.cov3D_precomp[(size_t)idx * 6 + k];
		} else {
			const float sc[3] = {a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]};
			const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
			cov3d_from_scale_rot(sc, a.scale_modifier, q, cov3D);
		}

		Cov2DTerms tt;
		cov2d_terms(p, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, vm, tt);
		float ca, cb, cc;
		cov2d_eval(tt, cov3D, ca, cb, cc);

		const float det = (ca * cc - cb * cb);
		if (det != 0.0f) {
			const float det_inv = 1.f / det;
			conic_a = cc * det_inv;
			conic_b = -cb * det_inv;
			conic_c = ca * det_inv;
			const float mid = 0.5f * (ca + cc);
			const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
			const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
			const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
			pix_x = ndc2pix(projx, a.W);
			pix_y = ndc2pix(projy, a.H);
			get_rect(pix_x, pix_y, (int)my_radius, a.gx, a.gy, rmin, rmax);
			if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) != 0) {
				radius_out = (int)my_radius;
				vis = true;
			}
		}
	}
	if (in_range && a.radii) a.radii[(size_t)view * a.P + idx] = radius_out;
	if (FILTER_ONLY) return;
	// rows without a rect need nothing else (readers look at the rect first); the padding rows between the views of a
	// batched call are rows too
	if (!vis && (in_range || a.n_views > 1)) a.geom.rect[id] = make_ushort4(0, 0, 0, 0);

	// ---- re-pack.  Part 2 (SH evaluation, the exact tile cull, the record) is 70 % of the kernel's instructions and a
	// wave runs it as long as ONE of its lanes is visible: on a camera sweep ~5 % of the Gaussians are, spread over
	// every wave, and the kernel was bound by exactly that divergence.  When the workgroup's visible Gaussians fit
	// fewer waves than they occupy, they are handed (through LDS, in id order) to the first threads of the workgroup:
	// thread t continues as the t-th visible Gaussian `li` of the workgroup, the other waves skip part 2.  Every value
	// is computed by the same instructions as before, only on another lane; the instance numbering below follows the
	// thread order, which is still the id order.
	int li = (int)threadIdx.x;   // Gaussian of the workgroup this thread works for from here on
	{
		constexpr int PACK_MAX = 128, NF = 21;
		__shared__ uint32_t s_pack[NF][PACK_MAX];
		__shared__ uint32_t s_nvis[4];
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		const uint64_t vm_mask = wave_ballot(vis);
		if (lane == 0) s_nvis[wave] = (uint32_t)__popcll(vm_mask);
		__syncthreads();
		const uint32_t v0 = s_nvis[0], v1 = s_nvis[1], v2 = s_nvis[2], v3 = s_nvis[3];
		const int n_vis = (int)(v0 + v1 + v2 + v3);
		const int waves_now = (v0 != 0) + (v1 != 0) + (v2 != 0) + (v3 != 0);
		if (n_vis <= PACK_MAX && ((n_vis + 63) >> 6) < waves_now) {   // workgroup-uniform
			if (vis) {
				const int slot = (int)((wave > 0 ? v0 : 0u) + (wave > 1 ? v1 : 0u) + (wave > 2 ? v2 : 0u)) +
				                 __popcll(vm_mask & ((1ull << lane) - 1ull));
				s_pack[0][slot] = (uint32_t)threadIdx.x;
				s_pack[1][slot] = __float_as_uint(p.x); s_pack[2][slot] = __float_as_uint(p.y); s_pack[3][slot] = __float_as_uint(p.z);
#pragma unroll
				for (int k = 0; k < 6; k++) s_pack[4 + k][slot] = __float_as_uint(cov3D[k]);
				s_pack[10][slot] = __float_as_uint(conic_a); s_pack[11][slot] = __float_as_uint(conic_b); s_pack[12][slot] = __float_as_uint(conic_c);
				s_pack[13][slot] = (uint32_t)radius_out;
				s_pack[14][slot] = __float_as_uint(pix_x); s_pack[15][slot] = __float_as_uint(pix_y);
				s_pack[16][slot] = (uint32_t)rmin[0]; s_pack[17][slot] = (uint32_t)rmin[1];
				s_pack[18][slot] = (uint32_t)rmax[0]; s_pack[19][slot] = (uint32_t)rmax[1];
				s_pack[20][slot] = __float_as_uint(pvz);
			}
			__syncthreads();
			const int t = (int)threadIdx.x;
			vis = t < n_vis;
			radius_out = 0;
			if (vis) {
				li = (int)s_pack[0][t];
				p = make_float3(__uint_as_float(s_pack[1][t]), __uint_as_float(s_pack[2][t]), __uint_as_float(s_pack[3][t]));
#pragma unroll
				for (int k = 0; k < 6; k++) cov3D[k] = __uint_as_float(s_pack[4 + k][t]);
				conic_a = __uint_as_float(s_pack[10][t]); conic_b = __uint_as_float(s_pack[11][t]); conic_c = __uint_as_float(s_pack[12][t]);
				radius_out = (int)s_pack[13][t];
				pix_x = __uint_as_float(s_pack[14][t]); pix_y = __uint_as_float(s_pack[15][t]);
				rmin[0] = (int)s_pack[16][t]; rmin[1] = (int)s_pack[17][t];
				rmax[0] = (int)s_pack[18][t]; rmax[1] = (int)s_pack[19][t];
				pvz = __uint_as_float(s_pack[20][t]);
			}
		}
	}
	const int gi = (int)blockIdx.x * 256 + li;   // the Gaussian (index into the caller's arrays)
	const int gid = wg * 256 + li;               // its geometry row

	// ---- part 2, visible Gaussians only
	if (vis) {
		rect_out = make_ushort4((unsigned short)rmin[0], (unsigned short)(rmin[1] + ty_off),
		                        (unsigned short)rmax[0], (unsigned short)(rmax[1] + ty_off));
		// (The reference keeps cov3D for its backward, forward.cu:208-214.  Here k_preprocess_bwd forms it again from
		// scale and rotation, which it reads anyway, with the same function -- the same bits -- instead of 24 B written
		// here and read there per Gaussian; a precomputed covariance is the caller's array in both passes.)
		float rgb[3];
		uint8_t clamp_bits = 0;
		if (COLOR == 1) {
			sh3_to_rgb_stream(p, a.cam_pos + 3 * view, a.shs + (size_t)gi * 48, rgb, clamp_bits);
		} else if (COLOR == 0 && a.colors_precomp == nullptr) {
			sh_to_rgb(a.D, a.M, p, a.cam_pos + 3 * view, a.shs + (size_t)gi * a.M * 3, rgb, clamp_bits);
		} else {
			rgb[0] = a.colors_precomp[3 * gi];
			rgb[1] = a.colors_precomp[3 * gi + 1];
			rgb[2] = a.colors_precomp[3 * gi + 2];
		}
		const float opacity = a.opacities[gi];
		// alpha = min(0.99, o*exp(power)) < 1/255  <=>  power < -ln(255*o); keep a margin far
		// above the rounding of logf/expf so the cut only skips pairs the exact test rejects.
		const float power_cut = -logf(255.0f * opacity) - 1.0e-3f;
		// The record carries the conic pre-scaled: (-a/2, -b, -c/2).  Scaling by a power of two commutes with
		// every rounding, so the walks' power = (ha dx) dx + (hc dy) dy + (nb dx) dy is bit for bit the
		// reference's -0.5f * (a dx dx + c dy dy) - b dx dy (forward.cu:419) with one multiply less per pair.
		rq0 = make_float4(pix_x, pix_y, -0.5f * conic_a, -conic_b);
		rq1 = make_float4(-0.5f * conic_c, power_cut, opacity, pvz);
		a.geom.clamped[gid] = clamp_bits;
		// Keep one instance per tile of the rect the splat can actually reach: the reference
		// lists every tile of the bounding rect (rasterizer_impl.cu:88-108); tiles where
		// alpha < 1/255 everywhere only ever `continue` in its render loops, so dropping them
		// changes no pixel.  Rects of more than 64 tiles are kept whole (mask too short).
		const uint32_t area = (uint32_t)(rmax[0] - rmin[0]) * (uint32_t)(rmax[1] - rmin[1]);
		const bool pd = (conic_a > 0.0f) && (conic_c > 0.0f) && (conic_a * conic_c - conic_b * conic_b > 0.0f);
		const float rb_c = -conic_b / conic_c, rb_a = -conic_b / conic_a;
		if (area <= 64u) {
			// Axis-aligned bounding box of the region {alpha >= 1/255} = {q(d) <= t}, t = -(cut - slack):
			// |dx| <= sqrt(2 t c / det), |dy| <= sqrt(2 t a / det).  Tiles outside it are dropped
			// without the edge test; NaN / non-PD conics fall back to the whole rect.
			int x0 = rmin[0], x1 = rmax[0], y0 = rmin[1], y1 = rmax[1];
			const float tq = -(power_cut - (1.0e-3f + 1.0e-4f * fabsf(power_cut)));
			const float cdet = conic_a * conic_c - conic_b * conic_b;
			if (pd) {
				if (tq < 0.0f) {
					x1 = x0;   // opacity below 1/255: nothing is ever blended
				} else {
					const float hx = sqrtf(2.0f * tq * conic_c / cdet) * 1.0001f + 0.01f;
					const float hy = sqrtf(2.0f * tq * conic_a / cdet) * 1.0001f + 0.01f;
					if (hx == hx && hy == hy) {
						// tile x holds pixel centres [16x, 16x+15]
						x0 = max(x0, (int)ceilf((pix_x - hx - 15.0f) * (1.0f / BSR_TILE)));
						x1 = min(x1, (int)floorf((pix_x + hx) * (1.0f / BSR_TILE)) + 1);
						y0 = max(y0, (int)ceilf((pix_y - hy - 15.0f) * (1.0f / BSR_TILE)));
						y1 = min(y1, (int)floorf((pix_y + hy) * (1.0f / BSR_TILE)) + 1);
					}
				}
			}
			const int w = rmax[0] - rmin[0];
			for (int y = y0; y < y1; y++)
				for (int x = x0; x < x1; x++) {
#ifdef BSR_PRE_KNOCK_CULL   // (cost attribution only: every tile of the clipped rect is kept)
					if (true) {
#else
					if (box_may_hit<15>(pix_x, pix_y, conic_a, conic_b, conic_c, power_cut, rb_c, rb_a, pd,
					                    (float)(x * BSR_TILE), (float)(y * BSR_TILE))) {
#endif
						kept_mask |= (1ull << (uint32_t)((y - rmin[1]) * w + (x - rmin[0])));
						atomicAdd(&s_hist[(uint32_t)((y + ty_off) * a.gx + x) & 255u], 1u);
					}
				}
		} else {
			for (int y = rmin[1]; y < rmax[1]; y++)
				for (int x = rmin[0]; x < rmax[0]; x++)
					atomicAdd(&s_hist[(uint32_t)((y + ty_off) * a.gx + x) & 255u], 1u);
		}
		rq2 = make_float4(rgb[0], rgb[1], rgb[2], __uint_as_float((uint32_t)(kept_mask >> 32)));
	}
	{
		// Gaussian-major instance blocks for the backward's gather: exclusive scan of the per-Gaussian
		// tile counts inside the workgroup; the per-workgroup totals are prefix-summed by k_scans.  Blocks of
		// different workgroups land in arbitrary order; inside a workgroup they ascend with the id.
		__shared__ uint32_t s_wave[4];
		__shared__ uint32_t s_area[4];
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		const uint32_t area_all = (uint32_t)(rect_out.z - rect_out.x) * (uint32_t)(rect_out.w - rect_out.y);
		const uint32_t n_inst = area_all ? kept_count(area_all, kept_mask) : 0u;
		// inclusive wave scans of the kept-instance counts (-> instance blocks, relative to the
		// workgroup) and wave sums of the rect areas (-> the reference's num_rendered)
		// (on DPP row shifts: twelve __shfl steps are twelve dependent ds_bpermute round trips at the end of every wave)
		const uint32_t incl = wave_inclusive_sum_dpp(n_inst);
		const uint32_t area_sum = wave_inclusive_sum_dpp(area_all);   // (lane 63: the wave's total)
		if (lane == 63) {
			s_wave[wave] = incl;
			s_area[wave] = area_sum;
		}
		__syncthreads();
		const uint32_t w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
		{
			const int per = ((int)(gridDim.x * gridDim.y) + 7) >> 3;
			// (view-batched calls: hist1 was zero-filled and most workgroups of a sparse view have nothing to add)
			if (a.n_views <= 1 || s_hist[threadIdx.x] != 0u)
				a.geom.hist1[(size_t)threadIdx.x * (8 * per) + hist1_column(wg, per)] = s_hist[threadIdx.x];
		}
		if (threadIdx.x == 0) {   // no global atomics: k_scans prefix-sums these per-workgroup totals
			a.geom.wg_kept[wg] = w0 + w1 + w2 + w3;
			a.geom.wg_area[wg] = s_area[0] + s_area[1] + s_area[2] + s_area[3];
		}
		if (vis) {
			const uint32_t off = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u) + incl - n_inst;
			a.geom.rect[gid] = rect_out;
			a.geom.depth[gid] = rq1.w;
			a.geom.inst_offset[gid] = off;
			a.geom.kept_mask[gid] = kept_mask;
			{
				// the whole 64-B splat record in four back-to-back stores: each cache line is written once,
				// completely (written piecemeal across the kernel, partially filled lines were evicted and
				// re-written: 181 MB of HBM writes for 113 MB of data at C3)
				float4* rec = a.geom.rec + (size_t)gid * BSR_REC;
				rec[0] = rq0;
				rec[1] = rq1;
				rec[2] = rq2;
				rec[3] = make_float4(
				    __uint_as_float(off), __uint_as_float((uint32_t)rect_out.x | ((uint32_t)rect_out.y << 16)),
				    __uint_as_float((uint32_t)(rect_out.z - rect_out.x) | ((uint32_t)(rect_out.w - rect_out.y) << 16)),
				    __uint_as_float((uint32_t)kept_mask));
			}
		}
	}
}

__global__ void __launch_bounds__(256) k_mark_visible(int P, const float* __restrict__ means3D,
                                                      const float* __restrict__ vm, uint8_t* __restrict__ present)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	const float pvz = vm[2] * means3D[3 * idx] + vm[6] * means3D[3 * idx + 1] + vm[10] * means3D[3 * idx + 2] + vm[14];
	present[idx] = !(pvz <= BSR_NEAR);
}

// visible_filter for V cameras in one pass over the Gaussians: the mean and the
// view-independent 3-D covariance are loaded/built once, then every view repeats exactly the
// arithmetic of k_preprocess<true> (same helpers, same order), so radii[v] is bit-identical to a
// single-view call with camera v.
//
// A camera path sees a given Gaussian from few of its views, and the lanes of a wave (64 neighbouring ids) rarely
// agree on which, so running the full test under divergence would cost every wave the full price for every view.
// Two phases per wave instead:
//   A  per view, all lanes: near plane + a CONSERVATIVE bounding-square test (cameras wave-uniform -> scalar loads).
//      With W the view rotation, J the projection Jacobian, S the 3-D covariance:
//        cov2D = (J W) S (J W)^t + 0.3 I,   |J|_F^2 <= K / tz^2  (K = fx^2 (1 + limx^2) + fy^2 (1 + limy^2), the clamp of
//        tx/tz, ty/tz bounds J02, J12),  B := K |W|_F^2 |S|_F / tz^2 >= |(J W) S (J W)^t|_2 for ANY symmetric S,
//        lambda_max = mid + sqrt(max(0.1, ((a-c)/2)^2 + b^2)) = max(largest eigenvalue of cov2D, mid + sqrt(0.1))
//        <= B + 0.3 + 0.317 for any symmetric S, definite or not (the eigenvalues of the 2x2 block lie in [-B, B]: no
//        positive semi-definiteness is assumed; tests/test_filter_cull_bound_cpu.py sweeps rank-2 covariances with
//        eigenvalues +s, -s).  The test uses 1.5 B + 0.62 (the extra half B covers the fp32 evaluation of mid^2 - det
//        on large entries): radius = ceil(3 sqrt(lambda)) <= rb := 3.01 sqrt(1.5015 B + 1) + 2.
//      get_rect() is empty when px + r < 0 or px - r >= 16 gx (same in y): tested with rb and a slack of 2 px + 1e-5 |px|
//      for the float evaluation of ndc2pix; every compare is false for NaN, so doubtful lanes stay candidates.
//   B  the surviving (lane, view) pairs are queued in LDS and evaluated 64 at a time with the exact code: lane i takes
//      pair i, reads that Gaussian's mean/covariance from the wave's LDS copy and the camera with vector loads.
// Group mode (bsr_visible_filter_groups): views carry a group id (the rank that renders them) and the kernel writes
// group_mask[g][idx] = "some view of group g sees the Gaussian" (+ per-workgroup row counts) instead of the V radii per
// Gaussian.
struct FilterWaveLds {
	float g[9][64];        // mean (3) + 3-D covariance (6) of the wave's 64 Gaussians, component-major
	uint32_t q[128];       // pending (view << 6 | lane) pairs
	uint32_t seen[2][64];  // group bits 0..31 / 32..63 per Gaussian
};

__device__ __forceinline__ void filter_wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(256) k_visible_filter_views(int P, int V, const float* __restrict__ means3D,
                                                              const float* __restrict__ scales, float scale_modifier,
                                                              const float* __restrict__ rotations,
                                                              const float* __restrict__ cov3D_precomp,
                                                              const float* __restrict__ viewmatrices,
                                                              const float* __restrict__ projmatrices, int W, int H,
                                                              float focal_x, float focal_y, float tan_fovx,
                                                              float tan_fovy, int gx, int gy, int* __restrict__ radii,
                                                              const int* __restrict__ group_of_view, int n_groups,
                                                              uint8_t* __restrict__ group_mask,
                                                              uint32_t* __restrict__ wg_counts)
{
	__shared__ FilterWaveLds lds_all[4];
	FilterWaveLds& L = lds_all[threadIdx.x >> 6];
	const int lane = threadIdx.x & 63;
	const int idx = blockIdx.x * 256 + threadIdx.x;
	const int wave_base = idx - lane;
	const bool valid = idx < P;
	const bool groups_only = (radii == nullptr && group_mask != nullptr);

	float3 p = make_float3(0.f, 0.f, 0.f);
	float cov3D[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	if (valid) {
		p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
		if (cov3D_precomp != nullptr) {
#pragma unroll
			for (int k = 0; k < 6; k++) cov3D[k] = cov3D_precomp[(size_t)idx * 6 + k];
		} else {
			const float sc[3] = {scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]};
			const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
			cov3d_from_scale_rot(sc, scale_modifier, q, cov3D);
		}
	}
	L.g[0][lane] = p.x; L.g[1][lane] = p.y; L.g[2][lane] = p.z;
#pragma unroll
	for (int k = 0; k < 6; k++) L.g[3 + k][lane] = cov3D[k];
	L.seen[0][lane] = 0u; L.seen[1][lane] = 0u;
	// |S|_F (>= the spectral radius of any symmetric S)
	const float sF = sqrtf(cov3D[0] * cov3D[0] + cov3D[3] * cov3D[3] + cov3D[5] * cov3D[5] +
	                       2.0f * (cov3D[1] * cov3D[1] + cov3D[2] * cov3D[2] + cov3D[4] * cov3D[4]));
	const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
	const float K = focal_x * focal_x * (1.0f + limx * limx) + focal_y * focal_y * (1.0f + limy * limy);
	const float xmax = 16.0f * (float)gx, ymax = 16.0f * (float)gy;
	filter_wave_sync();

	// phase B on n queued pairs starting at q[base]
	auto drain = [&](uint32_t base, uint32_t n) {
		if ((uint32_t)lane < n) {
			const uint32_t e = L.q[base + lane];
			const int src = (int)(e & 63u), v = (int)(e >> 6);
			const float3 pp = make_float3(L.g[0][src], L.g[1][src], L.g[2][src]);
			float cc3[6];
#pragma unroll
			for (int k = 0; k < 6; k++) cc3[k] = L.g[3 + k][src];
			const float* vm = viewmatrices + 16 * v;
			const float* pm = projmatrices + 16 * v;
			int radius_out = 0;
			const float pvz = vm[2] * pp.x + vm[6] * pp.y + vm[10] * pp.z + vm[14];
			if (!(pvz <= BSR_NEAR)) {
				const float hx = pm[0] * pp.x + pm[4] * pp.y + pm[8] * pp.z + pm[12];
				const float hy = pm[1] * pp.x + pm[5] * pp.y + pm[9] * pp.z + pm[13];
				const float hw = pm[3] * pp.x + pm[7] * pp.y + pm[11] * pp.z + pm[15];
				const float p_w = 1.0f / (hw + 0.0000001f);
				const float projx = hx * p_w, projy = hy * p_w;
				Cov2DTerms tt;
				cov2d_terms(pp, focal_x, focal_y, tan_fovx, tan_fovy, vm, tt);
				float ca, cb, cc;
				cov2d_eval(tt, cc3, ca, cb, cc);
				const float det = (ca * cc - cb * cb);
				if (det != 0.0f) {
					const float mid = 0.5f * (ca + cc);
					const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
					const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
					const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
					const float pix_x = ndc2pix(projx, W), pix_y = ndc2pix(projy, H);
					int rmin[2], rmax[2];
					get_rect(pix_x, pix_y, (int)my_radius, gx, gy, rmin, rmax);
					if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) != 0) radius_out = (int)my_radius;
				}
			}
			if (radii != nullptr) radii[(size_t)v * P + wave_base + src] = radius_out;
			if (group_mask != nullptr && radius_out > 0) {
				const int g = group_of_view[v] & 63;   // (an id outside [0, 64) must not index past the wave's LDS words)
				atomicOr(&L.seen[g >> 5][src], 1u << (g & 31));
			}
		}
		filter_wave_sync();
	};

	uint32_t count = 0;   // wave-uniform
	for (int v = 0; v < V; v++) {
		const float* vm = viewmatrices + 16 * v;
		const float* pm = projmatrices + 16 * v;
		bool cand = valid;
		if (groups_only) {
			// a group some earlier view already marked needs no further test (bits land with a queue's delay: harmless)
			const int g = group_of_view[v] & 63;
			cand = cand && !((L.seen[g >> 5][lane] >> (g & 31)) & 1u);
		}
		const float pvz = vm[2] * p.x + vm[6] * p.y + vm[10] * p.z + vm[14];
		cand = cand && !(pvz <= BSR_NEAR);
		{
			const float hx = pm[0] * p.x + pm[4] * p.y + pm[8] * p.z + pm[12];
			const float hy = pm[1] * p.x + pm[5] * p.y + pm[9] * p.z + pm[13];
			const float hw = pm[3] * p.x + pm[7] * p.y + pm[11] * p.z + pm[15];
			const float p_w = __builtin_amdgcn_rcpf(hw + 0.0000001f);
			const float pxf = ((hx * p_w + 1.0f) * (float)W - 1.0f) * 0.5f;
			const float pyf = ((hy * p_w + 1.0f) * (float)H - 1.0f) * 0.5f;
			const float wF2 = vm[0] * vm[0] + vm[1] * vm[1] + vm[2] * vm[2] + vm[4] * vm[4] + vm[5] * vm[5] +
			                  vm[6] * vm[6] + vm[8] * vm[8] + vm[9] * vm[9] + vm[10] * vm[10];
			const float itz = __builtin_amdgcn_rcpf(pvz);
			const float B = (K * wF2) * sF * (itz * itz);
			const float rb = 3.01f * __builtin_amdgcn_sqrtf(1.5015f * B + 1.0f) + 2.0f;
			const float slx = 2.0f + 1.0e-5f * fabsf(pxf), sly = 2.0f + 1.0e-5f * fabsf(pyf);
			const bool outside = (pxf + rb < -slx) || (pxf - rb > xmax + slx) || (pyf + rb < -sly) || (pyf - rb > ymax + sly);
			cand = cand && !outside;
		}
		if (radii != nullptr && valid && !cand) radii[(size_t)v * P + idx] = 0;
		const unsigned long long m = __ballot(cand);
		if (m != 0ull) {
			if (cand) L.q[count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
				((uint32_t)v << 6) | (uint32_t)lane;
			count += (uint32_t)__popcll(m);
			filter_wave_sync();
			if (count >= 64u) {
				count -= 64u;
				drain(count, 64u);
			}
		}
	}
	if (count != 0u) drain(0u, count);

	if (group_mask != nullptr) {
		// per-workgroup row counts, summed by k_sum_group_counts (one atomic per wave and group on n_groups addresses
		// cost 0.17 ms per group at 1 M Gaussians)
		__shared__ uint32_t s_cnt[4][64];
		const unsigned long long seen = (unsigned long long)L.seen[0][lane] | ((unsigned long long)L.seen[1][lane] << 32);
		for (int g = 0; g < n_groups; g++) {
			const bool bit = valid && ((seen >> g) & 1ull);
			if (valid) group_mask[(size_t)g * P + idx] = (uint8_t)bit;
			if (wg_counts != nullptr) {
				const unsigned long long mb = __ballot(bit);
				if (lane == 0) s_cnt[threadIdx.x >> 6][g] = (uint32_t)__popcll(mb);
			}
		}
		if (wg_counts != nullptr) {
			__syncthreads();
			if ((int)threadIdx.x < n_groups)
				wg_counts[(size_t)threadIdx.x * gridDim.x + blockIdx.x] =
				    s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
		}
	}
}

// group_counts[g] = sum of the n_wg per-workgroup counts of group g (one workgroup per group)
__global__ void __launch_bounds__(256) k_sum_group_counts(int n_wg, const uint32_t* __restrict__ wg_counts,
                                                          uint32_t* __restrict__ group_counts)
{
	__shared__ uint32_t part[256];
	uint32_t acc = 0;
	for (int i = threadIdx.x; i < n_wg; i += 256) acc += wg_counts[(size_t)blockIdx.x * n_wg + i];
	part[threadIdx.x] = acc;
	__syncthreads();
	for (int h = 128; h > 0; h >>= 1) {
		if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
		__syncthreads();
	}
	if (threadIdx.x == 0) group_counts[blockIdx.x] = part[0];
}

void launch_visible_filter_views(int P, int V, const float* means3D, const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp, const float* viewmatrices,
                                 const float* projmatrices, int W, int H, float tan_fovx, float tan_fovy, int* radii,
                                 const int* group_of_view, int n_groups, uint8_t* group_mask, uint32_t* wg_counts,
                                 uint32_t* group_counts, hipStream_t s)
{
	// wg_counts: scratch uint32 [n_groups][ceil(P / 256)] (needed iff group_counts is wanted)
	const int n_wg = (P + 255) / 256;
	hipLaunchKernelGGL(k_visible_filter_views, dim3(n_wg), dim3(256), 0, s, P, V, means3D, scales,
	                   scale_modifier, rotations, cov3D_precomp, viewmatrices, projmatrices, W, H,
	                   W / (2.0f * tan_fovx), H / (2.0f * tan_fovy), tan_fovx, tan_fovy, (W + BSR_TILE - 1) / BSR_TILE,
	                   (H + BSR_TILE - 1) / BSR_TILE, radii, group_of_view, n_groups, group_mask,
	                   group_counts != nullptr ? wg_counts : nullptr);
	if (group_counts != nullptr && n_groups > 0)
		hipLaunchKernelGGL(k_sum_group_counts, dim3(n_groups), dim3(256), 0, s, n_wg, wg_counts, group_counts);
}

void launch_preprocess(const PreArgs& a, bool filter_only, hipStream_t s)
{
	const int blocks = (a.P + 255) / 256;
	if (filter_only)
		hipLaunchKernelGGL((k_preprocess<true, 0>), dim3(blocks), dim3(256), 0, s, a);
	else if (a.colors_precomp == nullptr && a.D == 3 && a.M == 16)
		hipLaunchKernelGGL((k_preprocess<false, 1>), dim3(blocks, a.n_views > 1 ? a.n_views : 1), dim3(256), 0, s, a);
	else if (a.colors_precomp != nullptr)
		hipLaunchKernelGGL((k_preprocess<false, 2>), dim3(blocks, a.n_views > 1 ? a.n_views : 1), dim3(256), 0, s, a);
	else
		hipLaunchKernelGGL((k_preprocess<false, 0>), dim3(blocks, a.n_views > 1 ? a.n_views : 1), dim3(256), 0, s, a);
}

void launch_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present, hipStream_t s)
{
	hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, vm, present);
}

}  // namespace bsr
