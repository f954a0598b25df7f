// Row gathers for the view-parallel paths (views.scatter_visible_gaussians, views.render_views_sharded(compact=True)):
// the rows idx[r] of up to eight per-Gaussian fp32 tensors in ONE pass, either laid side by side in one packed matrix
// (bsr_pack_rows; torch: one index_select per tensor, then a cat) or into one destination per tensor
// (bsr_gather_rows; torch: one index_select per tensor).
// HBM-bound: 8 B per gathered float; a wave copies whole rows, consecutive lanes consecutive floats of the row.
#include "common.h"

namespace bsr {

struct PackTable {
	const float* src[BSR_PACK_MAX_SRC];
	float* dst[BSR_PACK_MAX_SRC];      // where column begin[k] of row 0 lands
	int dst_stride[BSR_PACK_MAX_SRC];  // floats between consecutive rows of dst[k]
	int begin[BSR_PACK_MAX_SRC + 1];   // first column of source k in the concatenated row; begin[n] = floats per row
	int n;
};

__global__ void __launch_bounds__(256) k_pack_rows(int R, int P, PackTable t, const int64_t* __restrict__ idx, int idx_stride)
{
	const int row = t.begin[t.n];
	const int lane = threadIdx.x & 63;
	const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
	const int n_waves = (gridDim.x * 256) >> 6;
	for (int r = wave; r < R; r += n_waves) {
		const int64_t i = idx[(size_t)r * idx_stride];
		const bool ok = i >= 0 && i < (int64_t)P;   // a row number outside the tensors packs as zeros, never as a wild read
		for (int c = lane; c < row; c += 64) {
			int k = 0;
#pragma unroll
			for (int j = 1; j < BSR_PACK_MAX_SRC; j++) k += (j < t.n && c >= t.begin[j]) ? 1 : 0;
			const int w = t.begin[k + 1] - t.begin[k];
			t.dst[k][(size_t)r * t.dst_stride[k] + (c - t.begin[k])] = ok ? t.src[k][(size_t)i * w + (c - t.begin[k])] : 0.0f;
		}
	}
}

// dst_packed != nullptr: one [R, sum(widths)] matrix; otherwise dst_each[k] = [R, widths[k]]
void launch_pack_rows(int R, int P, int n_src, const float* const* src, const int* widths, const int64_t* idx,
                      int idx_stride, float* dst_packed, float* const* dst_each, hipStream_t s)
{
	PackTable t;
	t.n = n_src;
	t.begin[0] = 0;
	for (int k = 0; k < BSR_PACK_MAX_SRC; k++) {
		t.src[k] = k < n_src ? src[k] : nullptr;
		t.begin[k + 1] = t.begin[k] + (k < n_src ? widths[k] : 0);
	}
	for (int k = 0; k < BSR_PACK_MAX_SRC; k++) {
		t.dst[k] = k >= n_src ? nullptr : (dst_packed ? dst_packed + t.begin[k] : dst_each[k]);
		t.dst_stride[k] = k >= n_src ? 0 : (dst_packed ? t.begin[n_src] : widths[k]);
	}
	const int waves_wanted = R < 256 * 8 * 4 * 4 ? R : 256 * 8 * 4 * 4;   // <= 8 workgroups of 4 waves per CU, 4 rounds
	const int blocks = (waves_wanted + 3) / 4;
	hipLaunchKernelGGL(k_pack_rows, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, s, R, P, t, idx, idx_stride);
}

}  // namespace bsr
