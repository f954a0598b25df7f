// Per-Gaussian backward: dL/d(conic, mean2D, colour) -> dL/d(mean3D, cov3D, SH, scale, rotation).
//
// Fuses the reference's two kernels computeCov2DCUDA (cuda_rasterizer/backward.cu:144-274) and
// preprocessCUDA (:346-396, with computeColorFromSH :20-139 and computeCov3D :278-341) into one
// streaming pass: dL_dmean3D is assigned by the cov2D part and then incremented by the projection
// and SH parts in the reference's order, without the round trip through HBM.  It also performs the
// gather half of the atomic-free accumulation (see render_bwd.hip): the Gaussian's per-instance
// partial sums are adjacent rows of the Gaussian-major slab (one per kept tile of its rect, at
// wg_base[g / 256] + inst_offset[g]) and are added in that fixed order.  Every
// output row is written here (zeros for culled Gaussians), so callers need no 300 MB memset
// (rasterize_points.cu:154-162 zero-fills everything up front).
#include "common.h"
#include "sh.h"

namespace bsr {


// Store the N = 3*(DEG+1)^2 computed coefficient gradients of one Gaussian (compile-time register
// indices, so the array never goes to scratch memory) and zero the remaining 3*M - N floats.
template <int N>
__device__ __forceinline__ void store_sh_grad(float* __restrict__ dst, int M, const float* v)
{
	const int total = M * 3;
	if ((M & 3) == 0) {   // 12*M bytes: the block is 16-B aligned and a whole number of float4
		float4* d4 = reinterpret_cast<float4*>(dst);
		constexpr int NV = (N + 3) / 4;
#pragma unroll
		for (int i = 0; i < NV; i++) {
			float4 o;
			o.x = (i * 4 + 0 < N) ? v[i * 4 + 0] : 0.f;
			o.y = (i * 4 + 1 < N) ? v[i * 4 + 1] : 0.f;
			o.z = (i * 4 + 2 < N) ? v[i * 4 + 2] : 0.f;
			o.w = (i * 4 + 3 < N) ? v[i * 4 + 3] : 0.f;
			if (i * 4 < total) d4[i] = o;
		}
		for (int i = NV; i < (total >> 2); i++) d4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	} else {
#pragma unroll
		for (int i = 0; i < N; i++)
			if (i < total) dst[i] = v[i];
		for (int i = N; i < total; i++) dst[i] = 0.f;
	}
}

__device__ __forceinline__ void zero_sh_grad(float* __restrict__ dst, int M)
{
	const int total = M * 3;
	if ((M & 3) == 0) {
		float4* d4 = reinterpret_cast<float4*>(dst);
		for (int i = 0; i < (total >> 2); i++) d4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	} else {
		for (int i = 0; i < total; i++) dst[i] = 0.f;
	}
}

template <int DEG>
__device__ __forceinline__ void sh_bwd_deg(const BwdArgs& a, int idx, const float3 m, const float* dcolor, float* dmean)
{
	constexpr int NC = (DEG + 1) * (DEG + 1);
	const float ox = m.x - a.campos[0], oy = m.y - a.campos[1], oz = m.z - a.campos[2];
	const float len = sqrtf((ox * ox + oy * oy) + oz * oz);
	const float x = ox / len, y = oy / len, z = oz / len;
	const uint8_t cl = a.geom.clamped[idx];
	float dL_dRGB[3] = {dcolor[0], dcolor[1], dcolor[2]};
	dL_dRGB[0] *= (cl & 1) ? 0.f : 1.f;
	dL_dRGB[1] *= (cl & 2) ? 0.f : 1.f;
	dL_dRGB[2] *= (cl & 4) ? 0.f : 1.f;
	{   // phase 1: coefficient gradients out to HBM before the coefficients come in
		float dsh[NC * 3];
		sh_coef_grad<DEG>(x, y, z, dL_dRGB, dsh);
		store_sh_grad<NC * 3>(a.dL_dsh + (size_t)idx * a.M * 3, a.M, dsh);
	}
	__builtin_amdgcn_sched_barrier(0);
	float dL_ddir[3];
	{   // phase 2: view-direction gradient needs the coefficients
		float c[NC * 3];
		load_sh<NC>(a.shs + (size_t)idx * a.M * 3, a.M, c);
		sh_dir_grad<DEG>(c, x, y, z, dL_dRGB, dL_ddir);
	}
	// dnormvdv(dir_orig, dL_ddir), reference auxiliary.h:107-117
	const float sum2 = ox * ox + oy * oy + oz * oz;
	const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
	dmean[0] += ((+sum2 - ox * ox) * dL_ddir[0] - oy * ox * dL_ddir[1] - oz * ox * dL_ddir[2]) * invsum32;
	dmean[1] += (-ox * oy * dL_ddir[0] + (sum2 - oy * oy) * dL_ddir[1] - oz * oy * dL_ddir[2]) * invsum32;
	dmean[2] += (-ox * oz * dL_ddir[0] - oy * oz * dL_ddir[1] + (sum2 - oz * oz) * dL_ddir[2]) * invsum32;
}

// Wave-cooperative variant of sh_bwd_deg for M = 4 / 16 (ROW_F4 = 3 / 12): every lane of the wave takes part; dL_dsh
// rows (zeros for culled Gaussians) leave through an LDS tile as contiguous wave stores and the coefficients come in
// the same way -- in two halves of 32 Gaussians, through a tile of 32 rows (6.6 KB per wave at M = 16), and neither the
// 48 coefficient gradients nor the 48 coefficients are ever held in registers: the gradients are formed from 16 basis
// factors as their row is written (sh_coef_basis), the view-direction gradient reads the coefficients from the LDS row
// channel by channel (sh_dir_grad_channel).  Same products, same sums in the same order as sh_bwd_deg.  Until round 6 a
// 64-row tile and both 48-float sets in registers: 168 VGPRs + 53 KB of LDS = 3 waves per SIMD for a kernel that waits
// for loads in 76 % of its wave cycles; now 78 VGPRs + 27 KB = 6 waves (C3: 130 -> 109 us, the dense leg 187 -> 158).
template <int DEG, int ROW_F4>
__device__ __forceinline__ void sh_bwd_coop_half(const BwdArgs& a, int idx, bool visible, const float3 m, const float* dcolor,
                                                 float* dmean, ShTile<ROW_F4, 32>& tile, int lane, int g0, int n_valid)
{
	constexpr int NC = (DEG + 1) * (DEG + 1);
	constexpr int N = NC * 3;
	constexpr int STRIDE = ShTile<ROW_F4, 32>::STRIDE;
	const float ox = m.x - a.campos[0], oy = m.y - a.campos[1], oz = m.z - a.campos[2];
	const float len = sqrtf((ox * ox + oy * oy) + oz * oz);
	// culled lanes use direction 0 and gradient 0: their row is exactly zero
	const float x = visible ? ox / len : 0.f, y = visible ? oy / len : 0.f, z = visible ? oz / len : 0.f;
	const uint8_t cl = visible ? a.geom.clamped[idx] : (uint8_t)7;
	float dL_dRGB[3] = {0.f, 0.f, 0.f};
	if (visible) {   // the reference's dL_dRGB *= clamped ? 0 : 1 (backward.cu:32-34)
		dL_dRGB[0] = dcolor[0] * ((cl & 1) ? 0.f : 1.f);
		dL_dRGB[1] = dcolor[1] * ((cl & 2) ? 0.f : 1.f);
		dL_dRGB[2] = dcolor[2] * ((cl & 4) ? 0.f : 1.f);
	}
	const int half = lane >> 5;
	float4* const row = &tile.rows[(lane & 31) * STRIDE];
	// phase 1: coefficient gradients -> own LDS row -> contiguous global stores, 32 Gaussians at a time
	{
		float b[NC];
		sh_coef_basis<DEG>(x, y, z, b);
#pragma unroll
		for (int h = 0; h < 2; h++) {
			if (half == h) {
#pragma unroll
				for (int i = 0; i < ROW_F4; i++) {
					float4 o;
					o.x = (i * 4 + 0 < N) ? b[(i * 4 + 0 < N) ? (i * 4 + 0) / 3 : 0] * dL_dRGB[(i * 4 + 0) % 3] : 0.f;
					o.y = (i * 4 + 1 < N) ? b[(i * 4 + 1 < N) ? (i * 4 + 1) / 3 : 0] * dL_dRGB[(i * 4 + 1) % 3] : 0.f;
					o.z = (i * 4 + 2 < N) ? b[(i * 4 + 2 < N) ? (i * 4 + 2) / 3 : 0] * dL_dRGB[(i * 4 + 2) % 3] : 0.f;
					o.w = (i * 4 + 3 < N) ? b[(i * 4 + 3 < N) ? (i * 4 + 3) / 3 : 0] * dL_dRGB[(i * 4 + 3) % 3] : 0.f;
					row[i] = o;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			sh_tile_store<ROW_F4, 32>(tile, a.dL_dsh, g0 + 32 * h, min(32, max(0, n_valid - 32 * h)), lane);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
	}
	// phase 2: coefficients in, view-direction gradient
	const uint64_t vis_mask = __ballot(visible);
	if (vis_mask == 0ull) return;
	float dL_ddir[3] = {0.f, 0.f, 0.f};
#pragma unroll
	for (int h = 0; h < 2; h++) {
		if (((vis_mask >> (32 * h)) & 0xffffffffull) == 0ull) continue;   // (wave-uniform)
		sh_tile_load<ROW_F4, 32>(tile, a.shs, g0 + 32 * h, min(32, max(0, n_valid - 32 * h)), lane);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (visible && half == h) {
			const float* const c = reinterpret_cast<const float*>(row);
			float dx0, dy0, dz0, dx1, dy1, dz1, dx2, dy2, dz2;
			sh_dir_grad_channel<DEG, 0>(c, x, y, z, dx0, dy0, dz0);
			__builtin_amdgcn_sched_barrier(0);
			sh_dir_grad_channel<DEG, 1>(c, x, y, z, dx1, dy1, dz1);
			__builtin_amdgcn_sched_barrier(0);
			sh_dir_grad_channel<DEG, 2>(c, x, y, z, dx2, dy2, dz2);
			dL_ddir[0] = (dx0 * dL_dRGB[0] + dx1 * dL_dRGB[1]) + dx2 * dL_dRGB[2];
			dL_ddir[1] = (dy0 * dL_dRGB[0] + dy1 * dL_dRGB[1]) + dy2 * dL_dRGB[2];
			dL_ddir[2] = (dz0 * dL_dRGB[0] + dz1 * dL_dRGB[1]) + dz2 * dL_dRGB[2];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();   // the other half's coefficients overwrite the tile
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	if (visible) {
		const float sum2 = ox * ox + oy * oy + oz * oz;
		const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
		dmean[0] += ((+sum2 - ox * ox) * dL_ddir[0] - oy * ox * dL_ddir[1] - oz * ox * dL_ddir[2]) * invsum32;
		dmean[1] += (-ox * oy * dL_ddir[0] + (sum2 - oy * oy) * dL_ddir[1] - oz * oy * dL_ddir[2]) * invsum32;
		dmean[2] += (-ox * oz * dL_ddir[0] - oy * oz * dL_ddir[1] + (sum2 - oz * oz) * dL_ddir[2]) * invsum32;
	}
}

// ROW_F4 = 0: per-thread SH access (any M); 3 / 12: wave-cooperative LDS-transposed access (M = 4 / 16);
// -1: no SH at all (precomputed colours, BloomScene's call shape): without the SH code the kernel needs 40 % fewer
// registers and nothing spills
template <int ROW_F4>
__global__ void __launch_bounds__(256, (ROW_F4 == 12 ? 6 : 4)) k_preprocess_bwd(const BwdArgs a)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	const bool in_range = idx < a.P;
	// (no early exit: whole waves take part in the slab gather and in the cooperative SH part)
	// More instances kept than the R this call was handed: the forward was a BSR_FLAG_NO_READBACK call whose capacity was
	// too small.  Its binning, lists and slab do not exist -- nothing of them is read: every Gaussian is treated as
	// culled and dL_dmean3D is filled with NaN (the frame was NaN too; the thread's next forward reports the overflow).
	const bool overflow = __builtin_amdgcn_readfirstlane(*a.kept_ptr) > a.capacity;
	const ushort4 rc = (in_range && !overflow) ? a.geom.rect[idx] : make_ushort4(0, 0, 0, 0);
	const bool visible = in_range && !overflow && (a.radii ? (a.radii[idx] > 0) : (rc.z > rc.x && rc.w > rc.y));

	// ---- add the per-instance partial sums of this Gaussian (adjacent slab rows) in tile (row-major) order.
	// Replaces the reference's 9 float atomicAdds per (pixel, Gaussian) pair (backward.cu:537,574-583);
	// the order is fixed, so the per-Gaussian sums do not depend on scheduling.
	//
	// The blocks of the 64 Gaussians of a wave are adjacent too (k_preprocess numbers the kept instances of a workgroup
	// in id order), so the wave's rows are ONE contiguous run of the slab.  A thread walking its own rows issues
	// row-strided loads, one dependent round trip per row, and the wave runs as many trips as its longest block;
	// instead the wave copies the run into LDS with coalesced 16-byte loads (all in flight at once) and every thread
	// adds its rows from there, in the same order: same sums bit for bit.  The staging area is the SH tile, which is
	// not in use yet (its own few KB in the instantiations without one).
	constexpr int STAGE_F4 = (ROW_F4 > 0 && 32 * (ROW_F4 + 1) > 192) ? 32 * (ROW_F4 + 1) : 192;   // float4 per wave
	__shared__ float4 s_stage[4][STAGE_F4];   // ROW_F4 > 0: reinterpreted as ShTile<ROW_F4, 32> by the SH part below
	float g[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	float g_z = 0.f;   // dL/d(view z): written only by the depth-gradient variant of k_render_bwd (else 0)
	{
		// rows are tight: 9 floats (10 with the depth gradient), so a run starts on a 4-byte boundary only
		const uint32_t rowf = (uint32_t)slab_row_floats(a.depth_grad != 0);
		const uint32_t stage_rows = (uint32_t)(STAGE_F4 * 4) / rowf;
		const int lane = threadIdx.x & 63;
		float4* const stage = s_stage[threadIdx.x >> 6];
		const float* const stage_f = reinterpret_cast<const float*>(stage);
		const uint32_t area = (uint32_t)(rc.z - rc.x) * (uint32_t)(rc.w - rc.y);
		const uint32_t n_inst = area ? kept_count(area, a.geom.kept_mask[idx]) : 0u;
		const uint32_t off = n_inst ? a.geom.wg_kept[idx >> 8] + a.geom.inst_offset[idx] : 0u;
		const uint32_t incl = wave_inclusive_sum_dpp(n_inst);   // (DPP row shifts; six __shfl_up steps: six ds_bpermute round trips)
		const uint32_t excl = incl - n_inst;
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		const uint64_t has = wave_ballot(n_inst != 0u);
		if (has != 0ull) {   // wave-uniform
			// first row of the run: any lane with a block knows it (its block starts `excl` rows in)
			const uint32_t run0 = (uint32_t)__builtin_amdgcn_readlane((int)(off - excl), (int)__builtin_ctzll(has));
			const float* const src = reinterpret_cast<const float*>(a.slab) + (size_t)run0 * rowf;
			for (uint32_t c0 = 0; c0 < total; c0 += stage_rows) {
				const uint32_t n_rows = min(stage_rows, total - c0);
				const uint32_t n_f4 = (n_rows * rowf + 3u) >> 2;   // (the last one may read up to 12 B past the run: the slab
				                                                   //  carries 16 spare bytes, api.hip)
				const bsr_f32x4_a4* const s = reinterpret_cast<const bsr_f32x4_a4*>(src + (size_t)c0 * rowf);
				for (uint32_t i0 = 0; i0 < n_f4; i0 += 256) {   // four 1-KiB loads in flight per trip
					const uint32_t i = i0 + (uint32_t)lane;
					bsr_f32x4 v0, v1, v2, v3;
					if (i < n_f4) v0 = s[i];
					if (i + 64 < n_f4) v1 = s[i + 64];
					if (i + 128 < n_f4) v2 = s[i + 128];
					if (i + 192 < n_f4) v3 = s[i + 192];
					if (i < n_f4) stage[i] = make_float4(v0.x, v0.y, v0.z, v0.w);
					if (i + 64 < n_f4) stage[i + 64] = make_float4(v1.x, v1.y, v1.z, v1.w);
					if (i + 128 < n_f4) stage[i + 128] = make_float4(v2.x, v2.y, v2.z, v2.w);
					if (i + 192 < n_f4) stage[i + 192] = make_float4(v3.x, v3.y, v3.z, v3.w);
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				const uint32_t k0 = max(excl, c0), k1 = min(incl, c0 + n_rows);
				for (uint32_t k = k0; k < k1; k++) {
					const float* row = stage_f + (k - c0) * rowf;
					g[0] += row[0]; g[1] += row[1]; g[2] += row[2]; g[3] += row[3];
					g[4] += row[4]; g[5] += row[5]; g[6] += row[6]; g[7] += row[7];
					g[8] += row[8];
					if (a.depth_grad) g_z += row[9];
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();   // the next chunk / the SH part overwrites the staging area
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			}
		}
	}
	if (in_range) {
		a.dL_dmean2D[3 * idx] = g[0];
		a.dL_dmean2D[3 * idx + 1] = g[1];
		a.dL_dmean2D[3 * idx + 2] = 0.f;
		// dL_dconic, dL_dcolor and dL_dcov3D are intermediate results whenever SH / scale+rotation inputs are
		// used: a caller that does not want them passes NULL and saves their HBM writes (52 B per Gaussian)
		if (a.dL_dconic) reinterpret_cast<float4*>(a.dL_dconic)[idx] = make_float4(g[2], g[3], 0.f, g[4]);
		a.dL_dopacity[idx] = g[5];
		if (a.dL_dcolor) {
			a.dL_dcolor[3 * idx] = g[6];
			a.dL_dcolor[3 * idx + 1] = g[7];
			a.dL_dcolor[3 * idx + 2] = g[8];
		}
	}

	float dmean[3] = {0.f, 0.f, 0.f};
	float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	float dscale[3] = {0.f, 0.f, 0.f};
	float drot[4] = {0.f, 0.f, 0.f, 0.f};

	const int li = in_range ? idx : 0;
	const float3 m = make_float3(a.means3D[3 * li], a.means3D[3 * li + 1], a.means3D[3 * li + 2]);
	if (visible) {
		const float* vm = a.viewmatrix;

		// ------------------------------------------------ computeCov2DCUDA, backward.cu:144-274
		float V[6];
		float sc_in[3] = {0.f, 0.f, 0.f};                   // scale and rotation: read once, used here and by computeCov3D below
		float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
		if (a.cov3D_precomp) {
#pragma unroll
			for (int k = 0; k < 6; k++) V[k] = a.cov3D_precomp[(size_t)idx * 6 + k];
		} else {
			// the forward's cov3D again (same function, same operands: same bits) instead of 24 B of scratch per Gaussian
			sc_in[0] = a.scales[3 * idx]; sc_in[1] = a.scales[3 * idx + 1]; sc_in[2] = a.scales[3 * idx + 2];
			q_in = reinterpret_cast<const float4*>(a.rotations)[idx];
			cov3d_from_scale_rot(sc_in, a.scale_modifier, q_in, V);
		}
		const float dcx = g[2], dcy = g[3], dcw = g[4];

		// shared prologue (same expressions as the forward)
		float tx = vm[0] * m.x + vm[4] * m.y + vm[8] * m.z + vm[12];
		float ty = vm[1] * m.x + vm[5] * m.y + vm[9] * m.z + vm[13];
		const float tz_ = vm[2] * m.x + vm[6] * m.y + vm[10] * m.z + vm[14];
		const float limx = 1.3f * a.tan_fovx;
		const float limy = 1.3f * a.tan_fovy;
		const float txtz = tx / tz_;
		const float tytz = ty / tz_;
		tx = fminf(limx, fmaxf(-limx, txtz)) * tz_;
		ty = fminf(limy, fmaxf(-limy, tytz)) * tz_;
		const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
		const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
		const float h_x = a.focal_x, h_y = a.focal_y;
		const float J00 = h_x / tz_;
		const float J02 = -(h_x * tx) / (tz_ * tz_);
		const float J11 = h_y / tz_;
		const float J12 = -(h_y * ty) / (tz_ * tz_);
		const float T00 = vm[0] * J00 + vm[2] * J02, T01 = vm[4] * J00 + vm[6] * J02, T02 = vm[8] * J00 + vm[10] * J02;
		const float T10 = vm[1] * J11 + vm[2] * J12, T11 = vm[5] * J11 + vm[6] * J12, T12 = vm[9] * J11 + vm[10] * J12;
		// Vrk (symmetric): V00=V[0] V01=V[1] V02=V[2] V11=V[3] V12=V[4] V22=V[5]
		const float V00 = V[0], V01 = V[1], V02 = V[2], V11 = V[3], V12 = V[4], V22 = V[5];
		const float A00 = T00 * V00 + T01 * V01 + T02 * V02;
		const float A10 = T00 * V01 + T01 * V11 + T02 * V12;
		const float A20 = T00 * V02 + T01 * V12 + T02 * V22;
		const float A01 = T10 * V00 + T11 * V01 + T12 * V02;
		const float A11 = T10 * V01 + T11 * V11 + T12 * V12;
		const float A21 = T10 * V02 + T11 * V12 + T12 * V22;
		float ca = A00 * T00 + A10 * T01 + A20 * T02;
		const float cb = A01 * T00 + A11 * T01 + A21 * T02;
		float cc = A01 * T10 + A11 * T11 + A21 * T12;
		ca += 0.3f;
		cc += 0.3f;
		const float a_ = ca, b_ = cb, c_ = cc;
		const float denom = a_ * c_ - b_ * b_;
		float dL_da = 0, dL_db = 0, dL_dc = 0;
		const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
		if (denom2inv != 0) {
			dL_da = denom2inv * (-c_ * c_ * dcx + 2 * b_ * c_ * dcy + (denom - a_ * c_) * dcw);
			dL_dc = denom2inv * (-a_ * a_ * dcw + 2 * a_ * b_ * dcy + (denom - a_ * c_) * dcx);
			dL_db = denom2inv * 2 * (b_ * c_ * dcx - (denom + 2 * b_ * b_) * dcy + a_ * b_ * dcw);
			dcov[0] = (T00 * T00 * dL_da + T00 * T10 * dL_db + T10 * T10 * dL_dc);
			dcov[3] = (T01 * T01 * dL_da + T01 * T11 * dL_db + T11 * T11 * dL_dc);
			dcov[5] = (T02 * T02 * dL_da + T02 * T12 * dL_db + T12 * T12 * dL_dc);
			dcov[1] = 2 * T00 * T01 * dL_da + (T00 * T11 + T01 * T10) * dL_db + 2 * T10 * T11 * dL_dc;
			dcov[2] = 2 * T00 * T02 * dL_da + (T00 * T12 + T02 * T10) * dL_db + 2 * T10 * T12 * dL_dc;
			dcov[4] = 2 * T02 * T01 * dL_da + (T01 * T12 + T02 * T11) * dL_db + 2 * T11 * T12 * dL_dc;
		}
		const float dL_dT00 = 2 * (T00 * V00 + T01 * V01 + T02 * V02) * dL_da + (T10 * V00 + T11 * V01 + T12 * V02) * dL_db;
		const float dL_dT01 = 2 * (T00 * V01 + T01 * V11 + T02 * V12) * dL_da + (T10 * V01 + T11 * V11 + T12 * V12) * dL_db;
		const float dL_dT02 = 2 * (T00 * V02 + T01 * V12 + T02 * V22) * dL_da + (T10 * V02 + T11 * V12 + T12 * V22) * dL_db;
		const float dL_dT10 = 2 * (T10 * V00 + T11 * V01 + T12 * V02) * dL_dc + (T00 * V00 + T01 * V01 + T02 * V02) * dL_db;
		const float dL_dT11 = 2 * (T10 * V01 + T11 * V11 + T12 * V12) * dL_dc + (T00 * V01 + T01 * V11 + T02 * V12) * dL_db;
		const float dL_dT12 = 2 * (T10 * V02 + T11 * V12 + T12 * V22) * dL_dc + (T00 * V02 + T01 * V12 + T02 * V22) * dL_db;
		// W[0][k] = vm[4k], W[1][k] = vm[1+4k], W[2][k] = vm[2+4k]
		const float dL_dJ00 = vm[0] * dL_dT00 + vm[4] * dL_dT01 + vm[8] * dL_dT02;
		const float dL_dJ02 = vm[2] * dL_dT00 + vm[6] * dL_dT01 + vm[10] * dL_dT02;
		const float dL_dJ11 = vm[1] * dL_dT10 + vm[5] * dL_dT11 + vm[9] * dL_dT12;
		const float dL_dJ12 = vm[2] * dL_dT10 + vm[6] * dL_dT11 + vm[10] * dL_dT12;
		const float tz = 1.f / tz_;
		const float tz2 = tz * tz;
		const float tz3 = tz2 * tz;
		const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
		const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
		const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * tx) * tz3 * dL_dJ02 + (2 * h_y * ty) * tz3 * dL_dJ12;
		dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
		dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
		dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;

		// ------------------------------------------------ preprocessCUDA (backward), backward.cu:346-396
		const float* proj = a.projmatrix;
		const float m_homw = proj[3] * m.x + proj[7] * m.y + proj[11] * m.z + proj[15];
		const float m_w = 1.0f / (m_homw + 0.0000001f);
		const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
		const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
		const float g2x = g[0], g2y = g[1];
		const float pdx = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
		const float pdy = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
		const float pdz = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
		dmean[0] += pdx;
		dmean[1] += pdy;
		dmean[2] += pdz;

		if (ROW_F4 == 0 && a.shs) {
			if (a.D <= 0) sh_bwd_deg<0>(a, idx, m, &g[6], dmean);
			else if (a.D == 1) sh_bwd_deg<1>(a, idx, m, &g[6], dmean);
			else if (a.D == 2) sh_bwd_deg<2>(a, idx, m, &g[6], dmean);
			else sh_bwd_deg<3>(a, idx, m, &g[6], dmean);
		}

		if (a.scales) {
			// computeCov3D (backward), backward.cu:278-341
			const float4 q = q_in;
			const float r = q.x, x = q.y, y = q.z, z = q.w;
			const float s[3] = {a.scale_modifier * sc_in[0], a.scale_modifier * sc_in[1], a.scale_modifier * sc_in[2]};
			// Rm[c][k] = R[c][k] (glm column c, row k)
			const float Rm[3][3] = {
			    {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
			    {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
			    {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
			float M2[3][3];   // 2 * M, M[c][k] = s_k * R[c][k]
#pragma unroll
			for (int c = 0; c < 3; c++)
#pragma unroll
				for (int k = 0; k < 3; k++) M2[c][k] = 2.0f * (s[k] * Rm[c][k]);
			const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
			                        {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
			                        {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
			float dM[3][3];   // dL_dM[c][rr] = sum_k M2[k][rr] * dS[c][k]
#pragma unroll
			for (int c = 0; c < 3; c++)
#pragma unroll
				for (int rr = 0; rr < 3; rr++) dM[c][rr] = M2[0][rr] * dS[c][0] + M2[1][rr] * dS[c][1] + M2[2][rr] * dS[c][2];
			// dL_dscale_i = dot(Rt[i], dL_dMt[i]) = sum_j R[j][i] * dM[j][i]
#pragma unroll
			for (int i = 0; i < 3; i++) dscale[i] = (Rm[0][i] * dM[0][i] + Rm[1][i] * dM[1][i]) + Rm[2][i] * dM[2][i];
			float G[3][3];    // dL_dMt[i][j] * s_i = dM[j][i] * s_i
#pragma unroll
			for (int i = 0; i < 3; i++)
#pragma unroll
				for (int j = 0; j < 3; j++) G[i][j] = dM[j][i] * s[i];
			drot[0] = 2 * z * (G[0][1] - G[1][0]) + 2 * y * (G[2][0] - G[0][2]) + 2 * x * (G[1][2] - G[2][1]);
			drot[1] = 2 * y * (G[1][0] + G[0][1]) + 2 * z * (G[2][0] + G[0][2]) + 2 * r * (G[1][2] - G[2][1]) - 4 * x * (G[2][2] + G[1][1]);
			drot[2] = 2 * x * (G[1][0] + G[0][1]) + 2 * r * (G[2][0] - G[0][2]) + 2 * z * (G[1][2] + G[2][1]) - 4 * y * (G[2][2] + G[0][0]);
			drot[3] = 2 * r * (G[0][1] - G[1][0]) + 2 * x * (G[2][0] + G[0][2]) + 2 * y * (G[1][2] + G[2][1]) - 4 * z * (G[1][1] + G[0][0]);
		}
	} else if (ROW_F4 == 0 && in_range && a.shs && a.dL_dsh) {
		zero_sh_grad(a.dL_dsh + (size_t)idx * a.M * 3, a.M);
	}

	// the covariance-side gradients are final here: out they go, before the SH part (13 registers it need not carry)
	if (in_range) {
		if (a.dL_dcov3D) {
#pragma unroll
			for (int k = 0; k < 6; k++) a.dL_dcov3D[(size_t)idx * 6 + k] = dcov[k];
		}
		if (a.scales) {
			a.dL_dscale[3 * idx] = dscale[0];
			a.dL_dscale[3 * idx + 1] = dscale[1];
			a.dL_dscale[3 * idx + 2] = dscale[2];
			reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
		}
	}

	__builtin_amdgcn_sched_barrier(0);   // (nothing of the SH part is scheduled into the chain above: registers)
	if (ROW_F4 > 0) {
		// SH part for the whole wave at once (the reference adds it to dL_dmean after the projection
		// part, backward.cu:387-391; the cov3D part above does not touch dL_dmean, so the order holds)
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		const int g0 = blockIdx.x * 256 + wave * 64;
		const int n_valid = min(64, max(0, a.P - g0));
		constexpr int RF = ROW_F4 > 0 ? ROW_F4 : 1, RF12 = ROW_F4 >= 12 ? ROW_F4 : 12;
		static_assert(sizeof(ShTile<RF, 32>) <= sizeof(s_stage[0]), "the SH tile lives in the staging area");
		ShTile<RF, 32>& tile = *reinterpret_cast<ShTile<RF, 32>*>(&s_stage[wave][0]);
		if (a.D <= 0) sh_bwd_coop_half<0, RF>(a, idx, visible, m, &g[6], dmean, tile, lane, g0, n_valid);
		else if (a.D == 1) sh_bwd_coop_half<1, RF>(a, idx, visible, m, &g[6], dmean, tile, lane, g0, n_valid);
		else if (ROW_F4 < 12) { /* degree > 1 needs M >= 9: not reachable with M = 4 (checked by the host) */ }
		else if (a.D == 2) sh_bwd_coop_half<2, RF12>(a, idx, visible, m, &g[6], dmean, *reinterpret_cast<ShTile<RF12, 32>*>(&tile), lane, g0, n_valid);
		else sh_bwd_coop_half<3, RF12>(a, idx, visible, m, &g[6], dmean, *reinterpret_cast<ShTile<RF12, 32>*>(&tile), lane, g0, n_valid);
	}
	if (!in_range) return;

	if (a.depth_grad && visible) {   // z_view = vm[2] x + vm[6] y + vm[10] z + vm[14]
		dmean[0] += a.viewmatrix[2] * g_z;
		dmean[1] += a.viewmatrix[6] * g_z;
		dmean[2] += a.viewmatrix[10] * g_z;
	}
	if (overflow) dmean[0] = dmean[1] = dmean[2] = __builtin_nanf("");   // (see the top of the kernel)
	a.dL_dmean3D[3 * idx] = dmean[0];
	a.dL_dmean3D[3 * idx + 1] = dmean[1];
	a.dL_dmean3D[3 * idx + 2] = dmean[2];
}

void launch_preprocess_bwd(const BwdArgs& a, hipStream_t s)
{
	const dim3 grid((a.P + 255) / 256), block(256);
	if (a.shs && a.M == 16)
		hipLaunchKernelGGL(k_preprocess_bwd<12>, grid, block, 0, s, a);
	else if (a.shs && a.M == 4)
		hipLaunchKernelGGL(k_preprocess_bwd<3>, grid, block, 0, s, a);
	else if (!a.shs)
		hipLaunchKernelGGL(k_preprocess_bwd<-1>, grid, block, 0, s, a);
	else
		hipLaunchKernelGGL(k_preprocess_bwd<0>, grid, block, 0, s, a);
}

}  // namespace bsr
