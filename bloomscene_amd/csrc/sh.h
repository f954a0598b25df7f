// Spherical-harmonics colour evaluation and its backward, per Gaussian.
// Semantics: reference cuda_rasterizer/forward.cu:20-71 and backward.cu:20-139, constants
// auxiliary.h:22-39.  The [P][M][3] coefficient block of one Gaussian is contiguous (12*M bytes);
// it is pulled in with 16-byte loads when 12*M is a multiple of 16 (M = 4, 16).
#pragma once
#include "common.h"

namespace bsr {

__device__ static const float SH_C0 = 0.28209479177387814f;
__device__ static const float SH_C1 = 0.4886025119029199f;
__device__ static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

// Load the first n_coef coefficients (n_coef*3 floats) of one Gaussian into registers.
template <int NC>
__device__ __forceinline__ void load_sh(const float* __restrict__ sh, int M, float* c)
{
	if ((M & 3) == 0) {   // 12*M bytes per Gaussian is a multiple of 16 -> block is 16-B aligned
		const float4* s4 = reinterpret_cast<const float4*>(sh);
		constexpr int NV = (NC * 3 + 3) / 4;
#pragma unroll
		for (int i = 0; i < NV; i++) {
			if (i * 4 < M * 3) {
				const float4 v = s4[i];
				if (i * 4 + 0 < NC * 3) c[i * 4 + 0] = v.x;
				if (i * 4 + 1 < NC * 3) c[i * 4 + 1] = v.y;
				if (i * 4 + 2 < NC * 3) c[i * 4 + 2] = v.z;
				if (i * 4 + 3 < NC * 3) c[i * 4 + 3] = v.w;
			}
		}
	} else {
#pragma unroll
		for (int i = 0; i < NC * 3; i++) c[i] = sh[i];
	}
}

template <int DEG>
__device__ __forceinline__ void sh_eval(const float* c, float x, float y, float z, float* res)
{
#pragma unroll
	for (int ch = 0; ch < 3; ch++) {
#define SH(k) c[(k) * 3 + ch]
		float r = SH_C0 * SH(0);
		if (DEG > 0) {
			r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
			if (DEG > 1) {
				const float xx = x * x, yy = y * y, zz = z * z;
				const float xy = x * y, yz = y * z, xz = x * z;
				r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
				    SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
				if (DEG > 2) {
					r = r + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
					    SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
					    SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
					    SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
					    SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
				}
			}
		}
#undef SH
		res[ch] = r + 0.5f;
	}
}

// Degree 3 with M = 16 (the C3 shape), streamed: the 48 floats of a Gaussian come in as two groups of six
// 16-byte loads (three of four until the second half of round 6: sh_eval3_stream) and every group is consumed before
// the next one is requested, so 24 instead of 48 coefficient registers are live (k_preprocess with all 48: 92 VGPRs ->
// occupancy 5; this path: 62 -> 8).  The sums run in exactly the
// order of sh_eval<3> (term k of channel ch is added k-th): bit-identical results.
__device__ __forceinline__ float sh3_basis(int k, float x, float y, float z, float xx, float yy, float zz, float xy,
                                           float yz, float xz)
{
	switch (k) {
	case 0: return SH_C0;
	case 1: return SH_C1 * y;
	case 2: return SH_C1 * z;
	case 3: return SH_C1 * x;
	case 4: return SH_C2[0] * xy;
	case 5: return SH_C2[1] * yz;
	case 6: return SH_C2[2] * (2.0f * zz - xx - yy);
	case 7: return SH_C2[3] * xz;
	case 8: return SH_C2[4] * (xx - yy);
	case 9: return SH_C3[0] * y * (3.0f * xx - yy);
	case 10: return SH_C3[1] * xy * z;
	case 11: return SH_C3[2] * y * (4.0f * zz - xx - yy);
	case 12: return SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
	case 13: return SH_C3[4] * x * (4.0f * zz - xx - yy);
	case 14: return SH_C3[5] * z * (xx - yy);
	default: return SH_C3[6] * x * (xx - 3.0f * yy);
	}
}

__device__ __forceinline__ void sh_eval3_stream(const float* __restrict__ sh, float x, float y, float z, float* res)
{
	const float xx = x * x, yy = y * y, zz = z * z;
	const float xy = x * y, yz = y * z, xz = x * z;
	const float4* s4 = reinterpret_cast<const float4*>(sh);
	float r[3] = {0.f, 0.f, 0.f};
	// Twelve 16-byte loads per row in groups, each group consumed before the next is requested (all twelve at once: 92
	// VGPRs, 5 waves per SIMD instead of 8).  Two groups of six (62 VGPRs), measured on one box: 78.5 us against 83.9 for
	// three groups of four (55 VGPRs), 93.6 for four of three, 99.0 for six of two; unequal pairs 8 + 4 / 7 + 5 / 4 + 8:
	// 88.6 / 86.6 / 93.8; groups 0 and 1 of three in flight together: 96.3.  (The rows are 192 B, the cache lines 128: a
	// line is asked for by more than one group, and a nontemporal load costs the kernel half its speed.)
#ifndef BSR_SH_GROUP
#define BSR_SH_GROUP 6
#endif
	constexpr int GS = BSR_SH_GROUP;
	static_assert(12 % GS == 0, "twelve 16-byte loads");
#pragma unroll
	for (int g = 0; g < 12 / GS; g++) {
		// the next group's address is made to depend on this point of the sums, or the compiler hoists all twelve
		// loads to the top again
		if (g > 0) asm volatile("" : "+v"(s4), "+v"(r[0]), "+v"(r[1]), "+v"(r[2]));
		float v[4 * GS];
#pragma unroll
		for (int i = 0; i < GS; i++) {
			const float4 q = s4[GS * g + i];
			v[4 * i + 0] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
		}
#pragma unroll
		for (int j = 0; j < 4 * GS; j++) {
			const int p = 4 * GS * g + j, k = p / 3, ch = p % 3;
			const float b = sh3_basis(k, x, y, z, xx, yy, zz, xy, yz, xz);
			if (k == 0) r[ch] = b * v[j];
			else if (k == 1 || k == 3) r[ch] = r[ch] - b * v[j];
			else r[ch] = r[ch] + b * v[j];
		}
	}
#pragma unroll
	for (int ch = 0; ch < 3; ch++) res[ch] = r[ch] + 0.5f;
}

// forward: rgb = max(0, 0.5 + sum basis*coef); clamp_bits bit ch set where the raw value was < 0
__device__ __forceinline__ void sh_to_rgb(int deg, int M, const float3 pos, const float* __restrict__ campos,
                                          const float* __restrict__ sh, float* rgb, uint8_t& clamp_bits)
{
	float dx = pos.x - campos[0], dy = pos.y - campos[1], dz = pos.z - campos[2];
	const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
	dx = dx / len; dy = dy / len; dz = dz / len;
	float raw[3];
	if (deg <= 0) {
		float c[3];
		load_sh<1>(sh, M, c);
		sh_eval<0>(c, dx, dy, dz, raw);
	} else if (deg == 1) {
		float c[12];
		load_sh<4>(sh, M, c);
		sh_eval<1>(c, dx, dy, dz, raw);
	} else if (deg == 2) {
		float c[27];
		load_sh<9>(sh, M, c);
		sh_eval<2>(c, dx, dy, dz, raw);
	} else {
		float c[48];
		load_sh<16>(sh, M, c);
		sh_eval<3>(c, dx, dy, dz, raw);
	}
	clamp_bits = 0;
#pragma unroll
	for (int ch = 0; ch < 3; ch++) {
		if (raw[ch] < 0) clamp_bits |= (uint8_t)(1u << ch);
		rgb[ch] = fmaxf(raw[ch], 0.0f);
	}
}

// forward colour for the one shape the streamed evaluation serves: active degree 3, M = 16
__device__ __forceinline__ void sh3_to_rgb_stream(const float3 pos, const float* __restrict__ campos,
                                                  const float* __restrict__ sh, float* rgb, uint8_t& clamp_bits)
{
	float dx = pos.x - campos[0], dy = pos.y - campos[1], dz = pos.z - campos[2];
	const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
	dx = dx / len; dy = dy / len; dz = dz / len;
	float raw[3];
	sh_eval3_stream(sh, dx, dy, dz, raw);
	clamp_bits = 0;
#pragma unroll
	for (int ch = 0; ch < 3; ch++) {
		if (raw[ch] < 0) clamp_bits |= (uint8_t)(1u << ch);
		rgb[ch] = fmaxf(raw[ch], 0.0f);
	}
}

// backward (reference backward.cu:20-139), split in two so that the 3*(DEG+1)^2 coefficient
// gradients and the 3*(DEG+1)^2 coefficients are never live in registers at the same time:
//  sh_coef_grad: dL_dsh[k] = basis_k(dir) * dL_dRGB   (needs no coefficients)
//  sh_dir_grad : dL_ddir = (dRGBdx.dL_dRGB, dRGBdy.dL_dRGB, dRGBdz.dL_dRGB)
template <int DEG>
__device__ __forceinline__ void sh_coef_grad(float x, float y, float z, const float* dL_dRGB, float* dsh)
{
#define DSH(k, v)                                                    \
	do {                                                             \
		const float _v = (v);                                        \
		_Pragma("unroll") for (int ch = 0; ch < 3; ch++) dsh[(k) * 3 + ch] = _v * dL_dRGB[ch]; \
	} while (0)
	DSH(0, SH_C0);
	if (DEG > 0) {
		DSH(1, -SH_C1 * y);
		DSH(2, SH_C1 * z);
		DSH(3, -SH_C1 * x);
		if (DEG > 1) {
			const float xx = x * x, yy = y * y, zz = z * z;
			const float xy = x * y, yz = y * z, xz = x * z;
			DSH(4, SH_C2[0] * xy);
			DSH(5, SH_C2[1] * yz);
			DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
			DSH(7, SH_C2[3] * xz);
			DSH(8, SH_C2[4] * (xx - yy));
			if (DEG > 2) {
				DSH(9, SH_C3[0] * y * (3.f * xx - yy));
				DSH(10, SH_C3[1] * xy * z);
				DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
				DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
				DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy));
				DSH(14, SH_C3[5] * z * (xx - yy));
				DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
			}
		}
	}
#undef DSH
}

template <int DEG>
__device__ __forceinline__ void sh_dir_grad(const float* c, float x, float y, float z, const float* dL_dRGB,
                                            float* dL_ddir)
{
	float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SH(k) c[(k) * 3 + ch]
	if (DEG > 0) {
#pragma unroll
		for (int ch = 0; ch < 3; ch++) {
			dRGBdx[ch] = -SH_C1 * SH(3);
			dRGBdy[ch] = -SH_C1 * SH(1);
			dRGBdz[ch] = SH_C1 * SH(2);
		}
		if (DEG > 1) {
			const float xx = x * x, yy = y * y, zz = z * z;
			const float xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
			for (int ch = 0; ch < 3; ch++) {
				dRGBdx[ch] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
				dRGBdy[ch] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
				dRGBdz[ch] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
			}
			if (DEG > 2) {
#pragma unroll
				for (int ch = 0; ch < 3; ch++) {
					dRGBdx[ch] += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz +
					               SH_C3[2] * SH(11) * -2.f * xy + SH_C3[3] * SH(12) * -3.f * 2.f * xz +
					               SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14) * 2.f * xz +
					               SH_C3[6] * SH(15) * 3.f * (xx - yy));
					dRGBdy[ch] += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
					               SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
					               SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
					               SH_C3[6] * SH(15) * -3.f * 2.f * xy);
					dRGBdz[ch] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
					               SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
					               SH_C3[5] * SH(14) * (xx - yy));
				}
			}
		}
	}
#undef SH
	dL_ddir[0] = (dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1]) + dRGBdx[2] * dL_dRGB[2];
	dL_ddir[1] = (dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1]) + dRGBdy[2] * dL_dRGB[2];
	dL_ddir[2] = (dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1]) + dRGBdz[2] * dL_dRGB[2];
}

// The factors of sh_coef_grad alone: dL_dsh[k][ch] = b[k] * dL_dRGB[ch] (one multiply each, as DSH above), so a caller
// can form the 3 * (DEG + 1)^2 products when it writes them out and never holds them all (16 instead of 48 registers).
template <int DEG>
__device__ __forceinline__ void sh_coef_basis(float x, float y, float z, float* b)
{
	b[0] = SH_C0;
	if (DEG > 0) {
		b[1] = -SH_C1 * y;
		b[2] = SH_C1 * z;
		b[3] = -SH_C1 * x;
		if (DEG > 1) {
			const float xx = x * x, yy = y * y, zz = z * z;
			const float xy = x * y, yz = y * z, xz = x * z;
			b[4] = SH_C2[0] * xy;
			b[5] = SH_C2[1] * yz;
			b[6] = SH_C2[2] * (2.f * zz - xx - yy);
			b[7] = SH_C2[3] * xz;
			b[8] = SH_C2[4] * (xx - yy);
			if (DEG > 2) {
				b[9] = SH_C3[0] * y * (3.f * xx - yy);
				b[10] = SH_C3[1] * xy * z;
				b[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
				b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
				b[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
				b[14] = SH_C3[5] * z * (xx - yy);
				b[15] = SH_C3[6] * x * (xx - 3.f * yy);
			}
		}
	}
}

// sh_dir_grad for ONE colour channel, its coefficients read where they lie (c[k * 3 + CH]: an LDS row): the same
// expressions in the same order as sh_dir_grad, so that a caller looping over the channels holds 15 coefficients at a
// time instead of 45.
template <int DEG, int CH>
__device__ __forceinline__ void sh_dir_grad_channel(const float* c, float x, float y, float z, float& dRGBdx, float& dRGBdy,
                                                    float& dRGBdz)
{
	dRGBdx = 0.f; dRGBdy = 0.f; dRGBdz = 0.f;
#define SH(k) c[(k) * 3 + CH]
	if (DEG > 0) {
		dRGBdx = -SH_C1 * SH(3);
		dRGBdy = -SH_C1 * SH(1);
		dRGBdz = SH_C1 * SH(2);
		if (DEG > 1) {
			const float xx = x * x, yy = y * y, zz = z * z;
			const float xy = x * y, yz = y * z, xz = x * z;
			dRGBdx += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
			dRGBdy += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
			dRGBdz += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
			if (DEG > 2) {
				dRGBdx += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz +
				           SH_C3[2] * SH(11) * -2.f * xy + SH_C3[3] * SH(12) * -3.f * 2.f * xz +
				           SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14) * 2.f * xz +
				           SH_C3[6] * SH(15) * 3.f * (xx - yy));
				dRGBdy += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
				           SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
				           SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
				           SH_C3[6] * SH(15) * -3.f * 2.f * xy);
				dRGBdz += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
				           SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
				           SH_C3[5] * SH(14) * (xx - yy));
			}
		}
	}
#undef SH
}

// ---- wave-cooperative, LDS-transposed access to the [P][M][3] coefficient arrays ---------------
// A lane reading or writing its own 12*M-byte block touches 64 different cache lines per wave
// instruction with 16 B each.  Instead the wave's 64 blocks (contiguous in memory, 12 KiB at M = 16)
// move with full 1-KiB wave instructions and are transposed through a padded LDS tile, one row
// per Gaussian.  ROW_F4 = 3*M/4 float4 per Gaussian (3 for M = 4, 12 for M = 16); the row stride is
// ROW_F4 + 1 float4 (52 dwords at M = 16: conflict-free for ds_read/write_b128).
template <int ROW_F4, int ROWS = 64>
struct ShTile {
	static constexpr int STRIDE = ROW_F4 + 1;
	float4 rows[ROWS * STRIDE];
};

// global -> LDS rows of the wave's Gaussians [g0, g0 + n_valid)
template <int ROW_F4, int ROWS>
__device__ __forceinline__ void sh_tile_load(ShTile<ROW_F4, ROWS>& t, const float* __restrict__ base, int g0, int n_valid,
                                             int lane)
{
	const float4* src = reinterpret_cast<const float4*>(base) + (size_t)g0 * ROW_F4;
	const int n_f4 = n_valid * ROW_F4;
#pragma unroll
	for (int i = 0; i < (ROW_F4 * ROWS + 63) / 64; i++) {
		const int n = i * 64 + lane;
		if (n < n_f4) {
			const int g = n / ROW_F4, part = n - g * ROW_F4;
			t.rows[g * ShTile<ROW_F4, ROWS>::STRIDE + part] = src[n];
		}
	}
}

// LDS rows -> global
template <int ROW_F4, int ROWS>
__device__ __forceinline__ void sh_tile_store(const ShTile<ROW_F4, ROWS>& t, float* __restrict__ base, int g0, int n_valid,
                                              int lane)
{
	float4* dst = reinterpret_cast<float4*>(base) + (size_t)g0 * ROW_F4;
	const int n_f4 = n_valid * ROW_F4;
#pragma unroll
	for (int i = 0; i < (ROW_F4 * ROWS + 63) / 64; i++) {
		const int n = i * 64 + lane;
		if (n < n_f4) {
			const int g = n / ROW_F4, part = n - g * ROW_F4;
			dst[n] = t.rows[g * ShTile<ROW_F4, ROWS>::STRIDE + part];
		}
	}
}

}  // namespace bsr
