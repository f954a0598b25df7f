// Backward tile renderer: back-to-front re-traversal producing dL/d(mean2D, conic, opacity, colour).
//
// Semantics: reference renderCUDA (backward), cuda_rasterizer/backward.cu:399-586 (under
// /root/reference/submodules/depth-diff-gaussian-rasterization); dL_depths is ignored exactly as
// there (:457-463,539-554 are commented out).  Per (pixel, Gaussian) pair the nine contributions
// are computed with the reference's operation order.
//
// The reference issues 9 lane-scattered float atomicAdds per pair.  On MI355X such atomics run
// ~17x below the contiguous rate, so the 9 partials are first summed over the wave's 64 pixels
// with DPP row operations (no LDS traffic), then over the tile's 4 waves with LDS atomics, and
// only one global atomic per (tile, Gaussian, component) is issued -- 256x fewer.
#include "common.h"

namespace bsr {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
	const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
	return v + __int_as_float(moved);
}

// Sum over the 64 lanes of the wave; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
	v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
	v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
	v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
	v = dpp_add<0x140, 0xf>(v);   // row_mirror      -> every lane holds its row-of-16 sum
	v = dpp_add<0x142, 0xa>(v);   // row_bcast15     -> rows 1,3 += rows 0,2
	v = dpp_add<0x143, 0xc>(v);   // row_bcast31     -> rows 2,3 += rows 0+1
	return v;
}

__global__ void __launch_bounds__(BSR_BLOCK) k_render_bwd(int n_tiles, int gx, int W, int H,
                                                          const uint32_t* __restrict__ tile_start,
                                                          const uint32_t* __restrict__ point_list,
                                                          const float4* __restrict__ rec,
                                                          const float* __restrict__ bg_color,
                                                          const float* __restrict__ final_Ts,
                                                          const uint32_t* __restrict__ n_contrib,
                                                          const float* __restrict__ dL_dpixels,
                                                          float* __restrict__ dL_dmean2D,   // [P,3]
                                                          float* __restrict__ dL_dconic2D,  // [P,4]
                                                          float* __restrict__ dL_dopacity,  // [P]
                                                          float* __restrict__ dL_dcolors)   // [P,3]
{
	__shared__ float4 s_q0[BSR_BLOCK];
	__shared__ float4 s_q1[BSR_BLOCK];
	__shared__ float4 s_q2[BSR_BLOCK];
	__shared__ uint32_t s_id[BSR_BLOCK];
	__shared__ float s_acc[9][BSR_BLOCK];
	__shared__ uint32_t s_touched[BSR_BLOCK];
	__shared__ uint32_t s_max[4];

	const int tile = xcd_tile(blockIdx.x, n_tiles);
	if (tile >= n_tiles) return;
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int tx = tile % gx, ty = tile / gx;
	const int px = tx * BSR_TILE + ((wave & 1) << 3) + (lane & 7);
	const int py = ty * BSR_TILE + ((wave >> 1) << 3) + (lane >> 3);
	const bool inside = px < W && py < H;
	const float pixfx = (float)px, pixfy = (float)py;
	const size_t pix_id = (size_t)W * py + px;
	const size_t plane = (size_t)H * W;

	const uint32_t start = tile_start[tile];

	const float T_final = inside ? final_Ts[pix_id] : 0.0f;
	float T = T_final;
	const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0u;
	float accum_rec0 = 0.f, accum_rec1 = 0.f, accum_rec2 = 0.f;
	float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f;
	if (inside) {
		dpx0 = dL_dpixels[pix_id];
		dpx1 = dL_dpixels[plane + pix_id];
		dpx2 = dL_dpixels[2 * plane + pix_id];
	}
	float last_alpha = 0.f;
	float last_c0 = 0.f, last_c1 = 0.f, last_c2 = 0.f;
	const float bg_dot_dpixel = bg_color[0] * dpx0 + bg_color[1] * dpx1 + bg_color[2] * dpx2;
	const float ddelx_dx = (float)(0.5 * W);
	const float ddely_dy = (float)(0.5 * H);

	// Entries at list positions >= max(last_contributor) are skipped by every pixel of the tile
	// (reference :498-500): start the walk at the deepest entry any pixel blended.
	uint32_t m = last_contributor;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
	if (lane == 0) s_max[wave] = m;
	s_acc[0][tid] = 0.f; s_acc[1][tid] = 0.f; s_acc[2][tid] = 0.f;
	s_acc[3][tid] = 0.f; s_acc[4][tid] = 0.f; s_acc[5][tid] = 0.f;
	s_acc[6][tid] = 0.f; s_acc[7][tid] = 0.f; s_acc[8][tid] = 0.f;
	s_touched[tid] = 0;
	__syncthreads();
	const int n_walk = (int)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));

	for (int base = 0; base < n_walk; base += BSR_BLOCK) {
		const int cnt = min(BSR_BLOCK, n_walk - base);
		const int top = n_walk - 1 - base;   // list position of batch entry j is top - j
		if (tid < cnt) {
			const uint32_t id = point_list[start + (uint32_t)(top - tid)];
			const float4* r = rec + (size_t)id * 3;
			s_id[tid] = id;
			s_q0[tid] = r[0];
			s_q1[tid] = r[1];
			s_q2[tid] = r[2];
		}
		__syncthreads();

		for (int j = 0; j < cnt; j++) {
			const uint32_t contributor = (uint32_t)(top - j);
			const float4 q0 = s_q0[j];
			const float4 q1 = s_q1[j];   // conic c, power cut, opacity, depth
			const float dx = q0.x - pixfx;
			const float dy = q0.y - pixfy;
			const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
			bool active = (contributor < last_contributor) && !(power > 0.0f) && !(power < q1.y);
			float G = 0.f, alpha = 0.f;
			if (active) {
				G = bsr_expf(power);
				alpha = fminf(0.99f, q1.z * G);
				active = !(alpha < 1.0f / 255.0f);
			}
			if (__ballot(active) == 0ull) continue;   // wave-uniform

			float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f;
			if (active) {
				const float4 q2 = s_q2[j];
				T = T / (1.f - alpha);
				const float dchannel_dcolor = alpha * T;
				float dL_dalpha = 0.0f;
				accum_rec0 = last_alpha * last_c0 + (1.f - last_alpha) * accum_rec0;
				last_c0 = q2.x;
				dL_dalpha += (q2.x - accum_rec0) * dpx0;
				v6 = dchannel_dcolor * dpx0;
				accum_rec1 = last_alpha * last_c1 + (1.f - last_alpha) * accum_rec1;
				last_c1 = q2.y;
				dL_dalpha += (q2.y - accum_rec1) * dpx1;
				v7 = dchannel_dcolor * dpx1;
				accum_rec2 = last_alpha * last_c2 + (1.f - last_alpha) * accum_rec2;
				last_c2 = q2.z;
				dL_dalpha += (q2.z - accum_rec2) * dpx2;
				v8 = dchannel_dcolor * dpx2;
				dL_dalpha *= T;
				last_alpha = alpha;
				dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
				const float dL_dG = q1.z * dL_dalpha;
				const float gdx = G * dx;
				const float gdy = G * dy;
				const float dG_ddelx = -gdx * q0.z - gdy * q0.w;
				const float dG_ddely = -gdy * q1.x - gdx * q0.w;
				v0 = dL_dG * dG_ddelx * ddelx_dx;
				v1 = dL_dG * dG_ddely * ddely_dy;
				v2 = -0.5f * gdx * dx * dL_dG;
				v3 = -0.5f * gdx * dy * dL_dG;
				v4 = -0.5f * gdy * dy * dL_dG;
				v5 = G * dL_dalpha;
			}
			v0 = wave_sum_to_lane63(v0);
			v1 = wave_sum_to_lane63(v1);
			v2 = wave_sum_to_lane63(v2);
			v3 = wave_sum_to_lane63(v3);
			v4 = wave_sum_to_lane63(v4);
			v5 = wave_sum_to_lane63(v5);
			v6 = wave_sum_to_lane63(v6);
			v7 = wave_sum_to_lane63(v7);
			v8 = wave_sum_to_lane63(v8);
			if (lane == 63) {
				atomicAdd(&s_acc[0][j], v0);
				atomicAdd(&s_acc[1][j], v1);
				atomicAdd(&s_acc[2][j], v2);
				atomicAdd(&s_acc[3][j], v3);
				atomicAdd(&s_acc[4][j], v4);
				atomicAdd(&s_acc[5][j], v5);
				atomicAdd(&s_acc[6][j], v6);
				atomicAdd(&s_acc[7][j], v7);
				atomicAdd(&s_acc[8][j], v8);
				s_touched[j] = 1;
			}
		}
		__syncthreads();
		if (tid < cnt && s_touched[tid]) {
			const uint32_t id = s_id[tid];
			atomicAdd(&dL_dmean2D[(size_t)id * 3 + 0], s_acc[0][tid]);
			atomicAdd(&dL_dmean2D[(size_t)id * 3 + 1], s_acc[1][tid]);
			atomicAdd(&dL_dconic2D[(size_t)id * 4 + 0], s_acc[2][tid]);
			atomicAdd(&dL_dconic2D[(size_t)id * 4 + 1], s_acc[3][tid]);
			atomicAdd(&dL_dconic2D[(size_t)id * 4 + 3], s_acc[4][tid]);
			atomicAdd(&dL_dopacity[id], s_acc[5][tid]);
			atomicAdd(&dL_dcolors[(size_t)id * 3 + 0], s_acc[6][tid]);
			atomicAdd(&dL_dcolors[(size_t)id * 3 + 1], s_acc[7][tid]);
			atomicAdd(&dL_dcolors[(size_t)id * 3 + 2], s_acc[8][tid]);
		}
		s_acc[0][tid] = 0.f; s_acc[1][tid] = 0.f; s_acc[2][tid] = 0.f;
		s_acc[3][tid] = 0.f; s_acc[4][tid] = 0.f; s_acc[5][tid] = 0.f;
		s_acc[6][tid] = 0.f; s_acc[7][tid] = 0.f; s_acc[8][tid] = 0.f;
		s_touched[tid] = 0;
		__syncthreads();
	}
}

void launch_render_bwd(int gx, int gy, int W, int H, const uint32_t* tile_start, const uint32_t* point_list,
                       const float4* rec, const float* bg, const float* final_T, const uint32_t* n_contrib,
                       const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                       hipStream_t s)
{
	const int n_tiles = gx * gy;
	const int blocks = ((n_tiles + 7) / 8) * 8;
	hipLaunchKernelGGL(k_render_bwd, dim3(blocks), dim3(BSR_BLOCK), 0, s, n_tiles, gx, W, H, tile_start, point_list, rec,
	                   bg, final_T, n_contrib, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor);
}

}  // namespace bsr
