// Per-tile binning and depth sort.
//
// The reference emits one 64-bit (tile | depth) key per (Gaussian, tile) instance in Gaussian order
// and runs a global stable radix sort over all R instances (rasterizer_impl.cu:70-111,304-309 under
// /root/reference/submodules/depth-diff-gaussian-rasterization), then finds tile ranges
// (:116-138).  The resulting order is (tile, depth bit pattern, Gaussian id).  Here the same
// order is produced without a global sort:
//   1. k_preprocess counted the instances of every tile (tile_count), dropping (Gaussian, tile)
//      pairs whose tile the splat provably cannot reach with alpha >= 1/255 (exact, conservative
//      ellipse-vs-tile test; ~1/3 of the reference's instances on the synthetic scenes): the
//      lists are order-preserving sub-sequences of the reference's and every pixel is unchanged,
//   2. k_scan_tiles turns the counts into tile ranges (one workgroup; <= a few 10k tiles),
//   3. k_scatter appends each instance's (depth_bits << 32 | id) key to its tile's segment
//      (atomic cursor -> arbitrary order inside the segment),
//   4. k_sort_tiles sorts each segment by that 64-bit key in LDS (bitonic network), which
//      restores exactly the stable-sort order because ids are unique within a tile.
// Traffic: 8 B written + 8 B read + 4 B written per instance instead of six 24-B radix passes.
#include "common.h"

namespace bsr {

// ---- exclusive scan of tile_count -> tile_start[0..T], total in tile_start[T]; zero cursors ----
// Exclusive scan of n values in place (dst may alias src), one 1024-thread workgroup; returns the total.
__device__ __forceinline__ uint32_t block_exclusive_scan(int n, const uint32_t* src, uint32_t* dst, uint32_t* s_wave,
                                                         uint32_t* s_carry, bool write)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid == 0) *s_carry = 0;
	__syncthreads();
	for (int base = 0; base < n; base += 1024) {
		const int i = base + tid;
		const uint32_t v = (i < n) ? src[i] : 0u;
		uint32_t incl = v;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(incl, d, 64);
			if (lane >= d) incl += t;
		}
		if (lane == 63) s_wave[wave] = incl;
		__syncthreads();
		uint32_t wave_off = 0;
		for (int w = 0; w < wave; w++) wave_off += s_wave[w];
		const uint32_t carry = *s_carry;
		if (write && i < n) dst[i] = carry + wave_off + incl - v;
		__syncthreads();
		if (tid == 1023) *s_carry = carry + wave_off + incl;
		__syncthreads();
	}
	return *s_carry;
}

__global__ void __launch_bounds__(1024) k_scan_tiles(int T, const uint32_t* __restrict__ tile_count,
                                                     uint32_t* __restrict__ tile_start,
                                                     uint32_t* __restrict__ tile_cursor, int n_wg,
                                                     uint32_t* __restrict__ wg_kept,
                                                     const uint32_t* __restrict__ wg_area, int* __restrict__ flags)
{
	{   // per-preprocess-workgroup totals: kept instances -> bases (in place), rect tiles -> num_rendered
		__shared__ uint32_t s_w[16];
		__shared__ uint32_t s_c;
		block_exclusive_scan(n_wg, wg_kept, wg_kept, s_w, &s_c, true);
		__syncthreads();
		const uint32_t total_area = block_exclusive_scan(n_wg, wg_area, nullptr, s_w, &s_c, false);
		if (threadIdx.x == 0) flags[3] = (int)total_area;
		__syncthreads();
	}
	__shared__ uint32_t s_wave[16];
	__shared__ uint32_t s_carry;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid == 0) s_carry = 0;
	__syncthreads();
	for (int base = 0; base < T; base += 1024) {
		const int i = base + tid;
		const uint32_t v = (i < T) ? tile_count[i] : 0u;
		uint32_t incl = v;   // inclusive wave scan
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t n = __shfl_up(incl, d, 64);
			if (lane >= d) incl += n;
		}
		if (lane == 63) s_wave[wave] = incl;
		__syncthreads();
		uint32_t wave_off = 0;
		for (int w = 0; w < wave; w++) wave_off += s_wave[w];
		const uint32_t carry = s_carry;
		if (i < T) {
			tile_start[i] = carry + wave_off + incl - v;
			tile_cursor[i] = 0;
		}
		__syncthreads();
		if (tid == 1023) s_carry = carry + wave_off + incl;
		__syncthreads();
	}
	if (tid == 0) tile_start[T] = s_carry;
}

// ---- append every instance to its tile's segment ----
__global__ void __launch_bounds__(256) k_scatter(int P, int gx, const ushort4* __restrict__ rect,
                                                 const uint64_t* __restrict__ kept_mask,
                                                 const float4* __restrict__ rec,
                                                 const uint32_t* __restrict__ tile_start,
                                                 uint32_t* __restrict__ tile_cursor, uint64_t* __restrict__ keys)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	const ushort4 r = rect[idx];
	if (r.z <= r.x || r.w <= r.y) return;
	const uint32_t depth_bits = __float_as_uint(rec[(size_t)idx * BSR_REC + 1].w);
	const uint64_t key = ((uint64_t)depth_bits << 32) | (uint32_t)idx;
	const uint32_t area = (uint32_t)(r.z - r.x) * (uint32_t)(r.w - r.y);
	const uint64_t mask = kept_mask[idx];
	uint32_t k = 0;
	for (int y = r.y; y < r.w; y++)
		for (int x = r.x; x < r.z; x++, k++) {
			if (!tile_kept(area, mask, k)) continue;   // same decision as the count in k_preprocess
			const int t = y * gx + x;
			const uint32_t pos = tile_start[t] + atomicAdd(&tile_cursor[t], 1u);
			keys[pos] = key;
		}
}

// ---- per-tile bitonic sort of 64-bit keys ----
// One workgroup per tile.  Segments of up to CAP keys are sorted in LDS; longer ones (rare: a
// tile overlapped by > CAP splats) are sorted in place in global memory by the same network.
// The network is the all-ascending form of bitonic sort (first step of every merge compares
// mirrored partners), so keys beyond n behave as +infinity pads without ever being stored:
// a compare-exchange whose upper index is >= n is a no-op.
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort_asc(KeyPtr k, int n, int tid)
{
	int n2 = 1;
	while (n2 < n) n2 <<= 1;
	for (int size = 2; size <= n2; size <<= 1) {
		const int half = size >> 1;
		__syncthreads();
		for (int i = tid; i < (n2 >> 1); i += BSR_BLOCK) {
			const int blk = i / half, off = i - blk * half;
			const int lo = blk * size + off;
			const int hi = blk * size + size - 1 - off;
			if (hi < n) {
				const uint64_t a = k[lo], b = k[hi];
				if (a > b) { k[lo] = b; k[hi] = a; }
			}
		}
		for (int stride = half >> 1; stride > 0; stride >>= 1) {
			__syncthreads();
			for (int i = tid; i < (n2 >> 1); i += BSR_BLOCK) {
				const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
				const int hi = lo | stride;
				if (hi < n) {
					const uint64_t a = k[lo], b = k[hi];
					if (a > b) { k[lo] = b; k[hi] = a; }
				}
			}
		}
	}
	__syncthreads();
}

template <int CAP>
__global__ void __launch_bounds__(BSR_BLOCK) k_sort_tiles(int T, int min_n, const uint32_t* __restrict__ tile_start,
                                                           const uint64_t* __restrict__ keys,
                                                           uint32_t* __restrict__ point_list)
{
	__shared__ uint64_t s_keys[CAP];
	const int tile = blockIdx.x;
	if (tile >= T) return;
	const uint32_t start = tile_start[tile];
	const int n = (int)(tile_start[tile + 1] - start);
	if (n <= min_n || n > CAP) return;   // handled by another size class
	const int tid = threadIdx.x;
	for (int i = tid; i < n; i += BSR_BLOCK) s_keys[i] = keys[start + i];
	bitonic_sort_asc(s_keys, n, tid);
	for (int i = tid; i < n; i += BSR_BLOCK) point_list[start + i] = (uint32_t)s_keys[i];
}

__global__ void __launch_bounds__(BSR_BLOCK) k_sort_tiles_global(int T, int min_n, const uint32_t* __restrict__ tile_start,
                                                                  uint64_t* keys, uint32_t* __restrict__ point_list)
{
	const int tile = blockIdx.x;
	if (tile >= T) return;
	const uint32_t start = tile_start[tile];
	const int n = (int)(tile_start[tile + 1] - start);
	if (n <= min_n) return;
	const int tid = threadIdx.x;
	bitonic_sort_asc(keys + start, n, tid);
	for (int i = tid; i < n; i += BSR_BLOCK) point_list[start + i] = (uint32_t)keys[start + i];
}

void launch_scan_tiles(int T, const uint32_t* tile_count, uint32_t* tile_start, uint32_t* tile_cursor, int n_wg,
                       uint32_t* wg_kept, const uint32_t* wg_area, int* flags, hipStream_t s)
{
	hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, T, tile_count, tile_start, tile_cursor, n_wg, wg_kept,
	                   wg_area, flags);
}

void launch_scatter(int P, int gx, const ushort4* rect, const uint64_t* kept_mask, const float4* rec,
                    const uint32_t* tile_start, uint32_t* tile_cursor, uint64_t* keys, hipStream_t s)
{
	hipLaunchKernelGGL(k_scatter, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, rect, kept_mask, rec, tile_start,
	                   tile_cursor, keys);
}

// Size classes: (0, 1024] -> 8 KB LDS, (1024, 8192] -> 64 KB LDS, > 8192 -> global memory.
void launch_sort_tiles(int T, int max_tile_hint, const uint32_t* tile_start, uint64_t* keys, uint32_t* point_list,
                       hipStream_t s)
{
	(void)max_tile_hint;
	hipLaunchKernelGGL(k_sort_tiles<1024>, dim3(T), dim3(BSR_BLOCK), 0, s, T, 0, tile_start, keys, point_list);
	hipLaunchKernelGGL(k_sort_tiles<8192>, dim3(T), dim3(BSR_BLOCK), 0, s, T, 1024, tile_start, keys, point_list);
	hipLaunchKernelGGL(k_sort_tiles_global, dim3(T), dim3(BSR_BLOCK), 0, s, T, 8192, tile_start, keys, point_list);
}

}  // namespace bsr
