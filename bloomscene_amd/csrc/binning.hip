// Per-tile binning and depth sort -- no floating-point atomics, and the few integer ones never decide a result.
//
// The reference emits one 64-bit (tile | depth) key per (Gaussian, tile) instance in Gaussian order
// and runs a global stable radix sort over all R instances on 32+bit key bits, six 8-bit passes at
// 1080p (rasterizer_impl.cu:70-111,304-309 under
// /root/reference/submodules/depth-diff-gaussian-rasterization), then finds tile ranges
// (:116-138).  The resulting order is (tile, depth bit pattern, Gaussian id).  Here:
//   1. k_preprocess decided per (Gaussian, tile) whether the splat can reach the tile at all
//      (exact conservative ellipse-vs-tile test; ~1/3 of the reference's instances are dropped on the
//      synthetic scenes, no pixel changes), numbered the kept instances inside its workgroup and
//      counted them per low tile-id byte (`hist1`, an LDS histogram per workgroup); k_scans
//      prefix-sums the workgroup totals and the 256 digit rows of hist1 (one launch),
//   2. k_emit_scatter writes every kept instance as one 12-byte element (tile id, Gaussian id,
//      depth bits) straight to its position after the FIRST radix pass: digit base + the prefix of
//      the earlier workgroups + an LDS counter.  The order inside one (workgroup, digit) group is
//      whatever the LDS atomics give -- it does not matter, see 5,
//   3. tile ids of up to 16 bits (every single-view call up to 4096 x 4096): the tile-owned second pass further down
//      (k_tile_count -> k_tile_starts -> k_tile_scatter), which also produces the tile ranges of 4.  Otherwise:
//      the remaining ceil(bits(T)/8) - 1 stable LSD radix passes on the TILE ID only (1080p: one
//      more pass; the reference does six over 45 key bits; elements move as single 12-byte
//      loads/stores): per-workgroup digit histogram -> 256 parallel row scans -> stable scatter
//      (wave ballots for the in-round rank, stamped per-wave counters across the 4 waves),
//   4. k_tile_ranges finds each tile's segment by a 16-ary search (one DPP row of 16 lanes per tile),
//   5. every segment is sorted by its 64-bit key in LDS (a bucket-and-rank sort where the depths spread, a bitonic
//      network where they pile up: rank_sort and the sections above it), which yields
//      exactly the reference's stable-sort order because ids are unique within a tile -- so the
//      order in which step 2 drops equal-tile elements never reaches the output,
//   6. frames of up to 8192 tiles with at most 1400 kept instances per tile take steps 3-5 in ONE launch
//      (k_bucket_sort: the tile segments laid out inside their pass-1 bucket).
// The earlier version counted and appended instances with global integer atomics (~20 G/s when
// lane-scattered on MI355X: 0.2 ms at C3, 1.4 ms at C5); this one is also fully deterministic.
#include "common.h"
#include <stdlib.h>

namespace bsr {

#define BSR_RADIX_BITS 8
#define BSR_RADIX_BINS 256
#define BSR_SORT_SMALL_N 1024   // tiles with more instances go to the wide sort classes (== BSR_SORT_SMALL below)

// Exclusive scan of n uint32 in place by ONE 1024-thread workgroup, 4 values per thread and step.
// Returns the total to every thread.  s_wave: 32 words (two sets of 16 wave totals, used in turn).
// One barrier per step: every thread adds up the sixteen wave totals itself (the carry lives in a register), and the
// totals of consecutive steps go to alternating sets, so that a step's writes are two barriers behind the last reads
// of the set they overwrite.  The next step's four values are requested before this step's barrier.  (Until round 6:
// three barriers per step, the carry through LDS, a step beginning with the wait for its own loads -- a row of C5's
// pass-1 histogram is five steps long: k_scans 34 -> .. us.)
struct AllValid { __device__ __forceinline__ bool operator()(int) const { return true; } };
template <typename Valid = AllValid>
__device__ __forceinline__ uint32_t block_exclusive_scan(int n, uint32_t* data, uint32_t* s_wave, bool write, Valid valid = Valid())
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint32_t carry = 0u;
	uint32_t nx[4];
#pragma unroll
	for (int k = 0; k < 4; k++) nx[k] = (tid * 4 + k < n && valid(tid * 4 + k)) ? data[tid * 4 + k] : 0u;
	int set = 0;
	for (int base = 0; base < n; base += 4096, set ^= 16) {
		const int i = base + tid * 4;
		uint32_t v[4];
#pragma unroll
		for (int k = 0; k < 4; k++) v[k] = nx[k];
#pragma unroll
		for (int k = 0; k < 4; k++) nx[k] = (i + 4096 + k < n && valid(i + 4096 + k)) ? data[i + 4096 + k] : 0u;
		const uint32_t mine = v[0] + v[1] + v[2] + v[3];
		const uint32_t incl = wave_inclusive_sum_dpp(mine);
		if (lane == 63) s_wave[set + wave] = incl;
		__syncthreads();
		uint32_t wave_off = 0u, total = 0u;
#pragma unroll
		for (int w = 0; w < 16; w++) {
			const uint32_t t = s_wave[set + w];
			wave_off += w < wave ? t : 0u;
			total += t;
		}
		uint32_t run = carry + wave_off + incl - mine;
		if (write) {
#pragma unroll
			for (int k = 0; k < 4; k++) {
				if (i + k < n) data[i + k] = run;
				run += v[k];
			}
		}
		carry += total;
	}
	__syncthreads();   // (the caller may reuse s_wave at once)
	return carry;
}

// The two scans between k_preprocess and the binning, in one launch:
//  workgroup 256     : per-preprocess-workgroup totals: kept instances -> workgroup bases (in place) and
//                      flags[2] = total kept; rect tiles -> flags[3] = the reference's num_rendered,
//  workgroups 0..255 : digit d's row of hist1 (the pass-1 histogram counted by k_preprocess, digit-major, one
//                      column per preprocess workgroup, XCD-grouped) -> exclusive prefixes in place + the digit
//                      total behind the rows.  Pad columns (no workgroup) are never written and count as 0.
struct Hist1ColumnValid {
	int per, n_wg;
	__device__ __forceinline__ bool operator()(int col) const { return hist1_wg_of_column(col, per) < n_wg; }
};
__global__ void __launch_bounds__(1024) k_scans(int n_wg, uint32_t* __restrict__ wg_kept,
                                                uint32_t* __restrict__ wg_area, int* __restrict__ flags,
                                                uint32_t* __restrict__ hist1, int* __restrict__ host_counts)
{
	__shared__ uint32_t s_w[32];
	if (blockIdx.x == BSR_RADIX_BINS) {
		const uint32_t kept = block_exclusive_scan(n_wg, wg_kept, s_w, true);
		const uint32_t area = block_exclusive_scan(n_wg, wg_area, s_w, false);
		if (threadIdx.x == 0) {
			flags[2] = (int)kept;
			flags[3] = (int)area;
			flags[1] = flags[4] = flags[5] = 0;   // counters of k_tile_ranges (the image buffer arrives uninitialised)
			flags[6] = flags[7] = 0;
			flags[BSR_POOL_FWD] = flags[BSR_POOL_BWD] = 0;   // pool counters of the two tile walks (common.h: pooled_tile)
			if (host_counts != nullptr) {
				// the forward's one read-back (reference rasterizer_impl.cu:282), without a copy: the four counters go
				// straight into the calling thread's pinned, device-mapped landing buffer; the host waits for the event
				// recorded behind this kernel (a 16-byte hipMemcpyAsync was a 4 us operation of its own on the stream)
				host_counts[0] = flags[0];
				host_counts[1] = 0;
				host_counts[2] = (int)kept;
				host_counts[3] = (int)area;
				__threadfence_system();
			}
		}
		return;
	}
	const int per = (n_wg + 7) >> 3, n_col = 8 * per;
	const uint32_t total = block_exclusive_scan(n_col, hist1 + (size_t)blockIdx.x * n_col, s_w, true,
	                                            Hist1ColumnValid{per, n_wg});
	if (threadIdx.x == 0) hist1[(size_t)BSR_RADIX_BINS * n_col + blockIdx.x] = total;   // digit totals behind the rows
}

// The binning kernels take the number of kept instances from DEVICE memory (flags[2], written by
// k_scans) and derive their work partition from it themselves, so the host can enqueue the whole
// binning stage before it has read that number back (bsr_forward overlaps its one blocking read with
// these kernels).  `capacity` is the number of instances the scratch buffers were sized for; if more
// were kept, every kernel returns at once and the host re-runs the stage with the right size.
struct RadixPartition { int n, chunk, n_blocks; };
__device__ __forceinline__ RadixPartition radix_partition(const int* __restrict__ n_ptr, int capacity,
                                                          int hist_blocks_max)
{
	RadixPartition p;
	p.n = *n_ptr;
	if (p.n > capacity || p.n < 0) p.n = -1;   // overflow: do nothing
	// chunk: multiple of 256, at least 1024 elements, at most hist_blocks_max workgroups
	int chunk = ((max(p.n, 0) + hist_blocks_max - 1) / hist_blocks_max + 255) / 256 * 256;
	p.chunk = chunk < 1024 ? 1024 : chunk;
	p.n_blocks = (max(p.n, 0) + p.chunk - 1) / p.chunk;
	return p;
}

// One workgroup per digit d: exclusive scan of row d of the digit-major histogram (in place) and the
// row total.  256 short, independent scans instead of one long latency-bound one.
__global__ void __launch_bounds__(256) k_radix_rowscan(const int* __restrict__ n_ptr, int capacity, int hist_blocks_max,
                                                       uint32_t* __restrict__ hist,
                                                       uint32_t* __restrict__ digit_total)
{
	__shared__ uint32_t s_wave[4];
	__shared__ uint32_t s_carry;
	const RadixPartition part = radix_partition(n_ptr, capacity, hist_blocks_max);
	if (part.n <= 0) return;
	const int n_blocks = part.n_blocks;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint32_t* row = hist + (size_t)blockIdx.x * n_blocks;
	if (tid == 0) s_carry = 0;
	__syncthreads();
	for (int base = 0; base < n_blocks; base += 256) {
		const int i = base + tid;
		const uint32_t v = (i < n_blocks) ? row[i] : 0u;
		uint32_t incl = v;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(incl, d, 64);
			if (lane >= d) incl += t;
		}
		if (lane == 63) s_wave[wave] = incl;
		__syncthreads();
		const uint32_t w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
		const uint32_t carry = s_carry;
		if (i < n_blocks) row[i] = carry + (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u) + incl - v;
		__syncthreads();
		if (tid == 0) s_carry = carry + w0 + w1 + w2 + w3;
		__syncthreads();
	}
	if (tid == 0) digit_total[blockIdx.x] = s_carry;
}

// ---- first radix pass, fused with the instance emit ----
// The per-workgroup histogram over the low 8 bits of the tile id was counted by k_preprocess (hist1,
// digit-major, one column per preprocess workgroup).  k_scans turns every digit's row into
// exclusive prefixes + the digit total; k_emit_scatter then writes each kept instance straight to its
// place in the pass-1 order: digit base + its workgroup's prefix + a running LDS counter.  The order
// inside a (workgroup, digit) bucket is arbitrary -- harmless, the per-tile sort orders by the unique
// (depth, id) key -- so no Gaussian-major staging array, no separate histogram pass.
__global__ void __launch_bounds__(256) k_emit_scatter(int P, int gx, const int* __restrict__ n_ptr, int capacity,
                                                      const ushort4* __restrict__ rect,
                                                      const uint64_t* __restrict__ kept_mask,
                                                      const float* __restrict__ depth,
                                                      const uint32_t* __restrict__ hist1, BinElem* __restrict__ elems,
                                                      uint32_t* __restrict__ zero_me, int n_zero, int compact)
{
	__shared__ uint32_t s_off[BSR_RADIX_BINS];   // next output position per digit for this workgroup
	__shared__ uint32_t s_scan[4];
	{
		const int n = *n_ptr;
		if (n <= 0 || n > capacity) return;
	}
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int n_wg = (int)gridDim.x, per = (n_wg + 7) >> 3, n_col = 8 * per;
	const int idx = blockIdx.x * 256 + tid;
	// counters of the tile-owned second pass (tile_count[T], tile_cursor[T]): zeroed here, one launch ahead of their use
	for (int i = idx; i < n_zero; i += n_wg * 256) zero_me[i] = 0u;
	// Everything this thread needs from global memory is requested up front: the kernel is a chain of dependent
	// round trips otherwise (totals -> barrier -> column -> barrier -> rect -> mask / depth), and loads do not move
	// across barriers on their own.  Rows of culled Gaussians hold an empty rect; their mask / depth are never
	// written, whatever is read there is not used.
	const bool in_range = idx < P;
	const int ld = in_range ? idx : 0;
	const uint32_t v = hist1[(size_t)BSR_RADIX_BINS * n_col + tid];
	const uint32_t col = hist1[(size_t)tid * n_col + hist1_column((int)blockIdx.x, per)];
	const ushort4 r = rect[ld];
	const uint64_t mask = kept_mask[ld];
	const uint32_t depth_bits = __float_as_uint(depth[ld]);
	{   // digit base = exclusive scan of the 256 digit totals (thread d <-> digit d) + this workgroup's prefix
		const uint32_t incl = wave_inclusive_sum_dpp(v);   // (six __shfl_up steps are six ds_bpermute round trips)
		if (lane == 63) s_scan[wave] = incl;
		__syncthreads();
		const uint32_t base = (wave > 0 ? s_scan[0] : 0u) + (wave > 1 ? s_scan[1] : 0u) + (wave > 2 ? s_scan[2] : 0u) + incl - v;
		s_off[tid] = base + col;
		// the 256 digit bases, once, behind the digit totals: k_bucket_sort takes its bucket's range from there instead
		// of every one of its workgroups scanning the totals again (a load, a scan and two barriers at its start)
		if (blockIdx.x == 0) const_cast<uint32_t*>(hist1)[(size_t)BSR_RADIX_BINS * n_col + BSR_RADIX_BINS + tid] = base;
	}
	__syncthreads();
	if (!in_range) return;
	if (r.z <= r.x || r.w <= r.y) return;
	const uint32_t area = (uint32_t)(r.z - r.x) * (uint32_t)(r.w - r.y);
	if (kept_count(area, mask) == 0) return;
	uint32_t k = 0;
	for (int y = r.y; y < r.w; y++)
		for (int x = r.x; x < r.z; x++, k++) {
			if (!tile_kept(area, mask, k)) continue;
			const uint32_t tile = (uint32_t)(y * gx + x);
			const uint32_t pos = atomicAdd(&s_off[tile & (BSR_RADIX_BINS - 1)], 1u);   // LDS
			store_elem_m(elems, pos, BinElem{tile, (uint32_t)idx, depth_bits}, compact);   // one 8- or 12-B store
		}
}

// ---- stable LSD radix pass on bits [shift, shift+8) of the tile id ----
// Workgroup b owns elements [b*chunk, (b+1)*chunk).  hist is digit-major: hist[d * n_blocks + b].
__global__ void __launch_bounds__(256) k_radix_hist(const int* __restrict__ n_ptr, int capacity, int hist_blocks_max,
                                                    int shift, const BinElem* __restrict__ elems,
                                                    uint32_t* __restrict__ hist)
{
	__shared__ uint32_t s_hist[BSR_RADIX_BINS];
	const RadixPartition part = radix_partition(n_ptr, capacity, hist_blocks_max);
	if ((int)blockIdx.x >= part.n_blocks) return;   // also the overflow / empty case (n_blocks = 0)
	const int n = part.n, chunk = part.chunk, n_blocks = part.n_blocks;
	const int tid = threadIdx.x;
	s_hist[tid] = 0;
	__syncthreads();
	const int beg = blockIdx.x * chunk, end = min(n, beg + chunk);
	for (int i = beg + tid; i < end; i += 1024) {   // four loads in flight per trip (the kernel is load latency)
		uint32_t t[4];
#pragma unroll
		for (int k = 0; k < 4; k++) t[k] = (i + 256 * k < end) ? elems[i + 256 * k].x : 0u;
#pragma unroll
		for (int k = 0; k < 4; k++)
			if (i + 256 * k < end) atomicAdd(&s_hist[(t[k] >> shift) & (BSR_RADIX_BINS - 1)], 1u);
	}
	__syncthreads();
	hist[(size_t)tid * n_blocks + blockIdx.x] = s_hist[tid];
}

__global__ void __launch_bounds__(256) k_radix_scatter(const int* __restrict__ n_ptr, int capacity,
                                                       int hist_blocks_max, int shift,
                                                       const BinElem* __restrict__ elems_in,
                                                       BinElem* __restrict__ elems_out,
                                                       const uint32_t* __restrict__ hist,
                                                       const uint32_t* __restrict__ digit_total)
{
	__shared__ uint32_t s_off[BSR_RADIX_BINS];        // next output position per digit for this workgroup
	__shared__ uint32_t s_wcnt[4][BSR_RADIX_BINS];    // (round << 8 | count) per wave and digit, stamped
	__shared__ uint32_t s_scan[4];
	const RadixPartition part = radix_partition(n_ptr, capacity, hist_blocks_max);
	if ((int)blockIdx.x >= part.n_blocks) return;
	const int n = part.n, chunk = part.chunk, n_blocks = part.n_blocks;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int beg = blockIdx.x * chunk, end = min(n, beg + chunk);
	// requested before the first barrier (loads do not move across barriers on their own): this block's row of
	// prefixes and the first round's element; inside the loop the next round's element is always in flight
	const uint32_t row_prefix = hist[(size_t)tid * n_blocks + blockIdx.x];
	BinElem e_next = BinElem{0u, 0u, 0u};
	if (beg + tid < end) e_next = load_elem(elems_in + beg + tid);
	{   // digit base = exclusive scan of the 256 digit totals (thread d <-> digit d) + this block's row prefix
		const uint32_t v = digit_total[tid];
		uint32_t incl = v;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(incl, d, 64);
			if (lane >= d) incl += t;
		}
		if (lane == 63) s_scan[wave] = incl;
		__syncthreads();
		const uint32_t base = (wave > 0 ? s_scan[0] : 0u) + (wave > 1 ? s_scan[1] : 0u) + (wave > 2 ? s_scan[2] : 0u) + incl - v;
		s_off[tid] = base + row_prefix;
	}
#pragma unroll
	for (int w = 0; w < 4; w++) s_wcnt[w][tid] = 0;
	__syncthreads();
	const unsigned long long lt = (1ull << lane) - 1ull;
	uint32_t round = 1;
	for (int base = beg; base < end; base += 256, round++) {
		const int i = base + tid;
		const bool valid = i < end;
		const BinElem e = e_next;
		if (i + 256 < end) e_next = load_elem(elems_in + i + 256);
		const uint32_t d = (e.x >> shift) & (BSR_RADIX_BINS - 1);
		// lanes of this wave with the same digit (invalid lanes match nobody)
		unsigned long long peers = __ballot(valid);
#pragma unroll
		for (int b = 0; b < BSR_RADIX_BITS; b++) {
			const bool bit = (d >> b) & 1u;
			const unsigned long long m = __ballot(bit);
			peers &= bit ? m : ~m;
		}
		const uint32_t rank_w = (uint32_t)__popcll(peers & lt);
		const uint32_t cnt_w = (uint32_t)__popcll(peers);
		if (valid && rank_w == 0) s_wcnt[wave][d] = (round << 8) | cnt_w;
		__syncthreads();
		uint32_t lower = 0;
		if (valid) {
#pragma unroll
			for (int w = 0; w < 4; w++) {
				const uint32_t c = s_wcnt[w][d];
				if (w < wave && (c >> 8) == round) lower += c & 0xffu;
			}
			const uint32_t pos = s_off[d] + lower + rank_w;
			store_elem(elems_out + pos, e);
		}
		__syncthreads();
		if (valid && rank_w == 0) atomicAdd(&s_off[d], cnt_w);   // LDS; order irrelevant, positions are taken
	}
}

// ---- second pass for tile ids of up to 16 bits: per-tile counts -> tile starts -> scatter to the tile's segment ----
// The order INSIDE a tile is free (the per-tile sort orders by the unique (depth, id) key), so the last pass needs
// neither a stable scatter nor a histogram per workgroup: after pass 1 the instances of tile t all sit in bucket
// t & 255, and
//   k_tile_count   : workgroups (bucket d1, slice s) count their slice by the high byte in LDS and add the non-zero
//                    bins to tile_count[t] (a few thousand global integer atomics in all),
//   k_tile_starts  : ONE workgroup turns the counts into tile_start[0 .. T] (exclusive scan in tile order) and files
//                    the tiles of the wide sort classes -- the 16-ary search of k_tile_ranges and its dependent
//                    loads through a 36 .. 180 MB array are gone,
//   k_tile_scatter : the same workgroups recount their slice, reserve a range in each tile they touch (one returning
//                    atomic per non-zero bin) and move their elements there (LDS cursors; no ballots, no barriers
//                    inside the loop).
// Against histogram + row scan + stable scatter + search (A/B on one box): binning C3 0.071 -> 0.064 ms, C5 0.333 ->
// 0.309, dense 0.193 -> 0.180, C2 0.022 -> 0.018.
struct BucketSlice { int beg, end; };
// [beg, end) of slice `s` (of n_slices) of pass-1 bucket d1; thread d holds digit d's total (256 threads)
__device__ __forceinline__ BucketSlice bucket_slice(uint32_t my_total, int d1, int s, int n_slices, uint32_t* s_scan,
                                                    uint32_t* s_base)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t incl = wave_inclusive_sum_dpp(my_total);
	if (lane == 63) s_scan[wave] = incl;
	__syncthreads();
	const uint32_t base = (wave > 0 ? s_scan[0] : 0u) + (wave > 1 ? s_scan[1] : 0u) + (wave > 2 ? s_scan[2] : 0u) + incl - my_total;
	if (tid == d1) {
		s_base[0] = base;
		s_base[1] = my_total;
	}
	__syncthreads();
	const uint32_t b0 = s_base[0], size = s_base[1];
	const uint32_t len = ((size + (uint32_t)n_slices - 1) / (uint32_t)n_slices + 255u) & ~255u;   // multiple of 256
	BucketSlice r;
	r.beg = (int)(b0 + min(size, (uint32_t)s * len));
	r.end = (int)(b0 + min(size, (uint32_t)(s + 1) * len));
	return r;
}

// LDS histogram of the high byte over elems[beg, end): four loads in flight per trip
__device__ __forceinline__ void slice_histogram(const BinElem* __restrict__ elems, BucketSlice sl, uint32_t* s_hist, int compact)
{
	const int tid = threadIdx.x;
	for (int i = sl.beg + tid; i < sl.end; i += 1024) {
		uint32_t t[4];
#pragma unroll
		for (int k = 0; k < 4; k++) t[k] = (i + 256 * k < sl.end) ? elem_tile_m(elems, (size_t)(i + 256 * k), compact) : 0u;
#pragma unroll
		for (int k = 0; k < 4; k++)
			if (i + 256 * k < sl.end) atomicAdd(&s_hist[(t[k] >> BSR_RADIX_BITS) & (BSR_RADIX_BINS - 1)], 1u);
	}
}

__global__ void __launch_bounds__(256) k_tile_count(int T, int n_slices, const int* __restrict__ n_ptr, int capacity,
                                                    const uint32_t* __restrict__ digit_total1,
                                                    const BinElem* __restrict__ elems, uint32_t* __restrict__ tile_count,
                                                    int compact)
{
	__shared__ uint32_t s_hist[BSR_RADIX_BINS];
	__shared__ uint32_t s_scan[4];
	__shared__ uint32_t s_base[2];
	{
		const int n = *n_ptr;
		if (n <= 0 || n > capacity) return;
	}
	const int tid = threadIdx.x;
	const int d1 = (int)blockIdx.x / n_slices, s = (int)blockIdx.x % n_slices;
	s_hist[tid] = 0;
	const BucketSlice sl = bucket_slice(digit_total1[tid], d1, s, n_slices, s_scan, s_base);   // (barriers inside)
	slice_histogram(elems, sl, s_hist, compact);
	__syncthreads();
	const uint32_t c = s_hist[tid];
	const uint32_t t = ((uint32_t)tid << BSR_RADIX_BITS) | (uint32_t)d1;
	if (c != 0u && t < (uint32_t)T) atomicAdd(&tile_count[t], c);
}

// ONE workgroup of 1024: tile_count -> tile_start (exclusive scan in tile order, tile_start[T] = total) and the work
// lists of the wide sort classes (as k_tile_ranges builds them).  (Folding this into k_tile_count's last-finishing
// workgroup was measured: the per-workgroup ordering it needs -- returning atomics + one contended counter -- cost 4x
// the launch it saves.)  The tiles are taken in chunks of 1024 consecutive ones, thread i <-> tile 1024 j + i: every
// access is coalesced (a thread owning `per` CONSECUTIVE tiles made each wave load touch 64 cache lines: 105 us on
// this one CU at 65 536 tiles -- 8 stacked 1080p views, one 8K view).  All counts are requested up front and stay in
// registers; a wave scans its 64 tiles of every chunk, then the 16 x (chunks) wave totals are scanned once.
template <int NC>   // chunks held in registers: 8 (up to 8192 tiles: one 1080p view) or 64 (16-bit tile ids)
__global__ void __launch_bounds__(1024) k_tile_starts(int T, const int* __restrict__ n_ptr, int capacity,
                                                      const uint32_t* __restrict__ tile_count,
                                                      uint2* __restrict__ tile_range,
                                                      uint32_t* __restrict__ big_tiles, int* __restrict__ flags)
{
	__shared__ uint32_t s_tot[1024];         // [chunk][wave] totals (NC x 16 used), then their exclusive prefix
	__shared__ uint32_t s_w[16];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int chunks = (T + 1023) >> 10;     // <= NC
	uint32_t c[NC];
#pragma unroll
	for (int j = 0; j < NC; j++) c[j] = (j < chunks && (j << 10) + tid < T) ? tile_count[(j << 10) + tid] : 0u;
	{
		const int n = *n_ptr;
		if (n > capacity) return;   // overflow: the stage is re-run
		if (n <= 0) {               // nothing kept: every tile is empty (the counts were not even zeroed)
			for (int t = tid; t < T; t += 1024) tile_range[t] = make_uint2(0u, 0u);
			return;
		}
	}
	// inclusive scan of every chunk's 64 tiles of this wave (c[j] becomes the inclusive sum, the count is re-derived)
	bool any_big = false;
	s_tot[tid] = 0u;
	__syncthreads();
#pragma unroll
	for (int j = 0; j < NC; j++) {
		if (j < chunks) {   // (uniform)
			any_big = any_big || c[j] > (uint32_t)BSR_SORT_SMALL_N;
			const uint32_t incl = wave_inclusive_sum_dpp(c[j]);
			if (lane == 63) s_tot[j * 16 + wave] = incl;
			c[j] = incl - c[j];   // exclusive within the wave
		}
	}
	__syncthreads();
	// exclusive scan of the 1024 (chunk, wave) totals: thread t owns total t
	{
		const uint32_t mine = s_tot[tid];
		const uint32_t incl = wave_inclusive_sum_dpp(mine);
		if (lane == 63) s_w[wave] = incl;
		__syncthreads();
		uint32_t run = incl - mine;
		for (int w = 0; w < wave; w++) run += s_w[w];
		s_tot[tid] = run;
	}
	__syncthreads();
#pragma unroll
	for (int j = 0; j < NC; j++)
		if (j < chunks && (j << 10) + tid < T) {
			const uint32_t first = s_tot[j * 16 + wave] + c[j];
			tile_range[(j << 10) + tid] = make_uint2(first, first + tile_count[(j << 10) + tid]);   // (the count: an L2 hit)
		}
	if (__syncthreads_or(any_big)) {   // the filing re-reads the counts
		// list positions from LDS counters (this is the only workgroup): a returning GLOBAL atomic per wave, chunk and
		// class was 8 us of this kernel's 12 on the dense leg, where most tiles are long
		__shared__ int s_cls[3];
		if (tid < 3) s_cls[tid] = 0;
		__syncthreads();
		for (int j = 0; j < chunks; j++) {
			const int t = (j << 10) + tid;
			const uint32_t cc = t < T ? tile_count[t] : 0u;
			const int cls = cc > 8192u ? 2 : (cc > 4096u ? 1 : (cc > (uint32_t)BSR_SORT_SMALL_N ? 0 : -1));
#pragma unroll
			for (int c3 = 0; c3 < 3; c3++) {
				const uint64_t b = wave_ballot(cls == c3);
				if (b == 0ull) continue;
				int base = 0;
				if (lane == 0) base = atomicAdd(&s_cls[c3], __popcll(b));   // LDS
				base = __shfl(base, 0);
				if (cls == c3) big_tiles[(size_t)c3 * T + base + __popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)t;
			}
		}
		__syncthreads();
		if (tid < 3) flags[tid == 0 ? 1 : 3 + tid] = s_cls[tid];   // (zero until now: k_scans / the re-run's memset)
	}
}

#define BSR_SLICE_REGS 16   // elements per thread held in registers by k_tile_scatter: slices of up to 4096 are read once
__global__ void __launch_bounds__(256) k_tile_scatter(int T, int n_slices, const int* __restrict__ n_ptr, int capacity,
                                                      const uint32_t* __restrict__ digit_total1,
                                                      const BinElem* __restrict__ elems_in,
                                                      BinElem* __restrict__ elems_out,
                                                      const uint2* __restrict__ tile_range,
                                                      uint32_t* __restrict__ tile_cursor, int compact)
{
	__shared__ uint32_t s_hist[BSR_RADIX_BINS];
	__shared__ uint32_t s_off[BSR_RADIX_BINS];
	__shared__ uint32_t s_scan[4];
	__shared__ uint32_t s_base[2];
	{
		const int n = *n_ptr;
		if (n <= 0 || n > capacity) return;
	}
	const int tid = threadIdx.x;
	const int d1 = (int)blockIdx.x / n_slices, s = (int)blockIdx.x % n_slices;
	s_hist[tid] = 0;
	const BucketSlice sl = bucket_slice(digit_total1[tid], d1, s, n_slices, s_scan, s_base);
	if (sl.beg >= sl.end) return;   // (uniform)
	const bool in_regs = sl.end - sl.beg <= 256 * BSR_SLICE_REGS;   // (uniform) the usual case
	BinElem e[BSR_SLICE_REGS];
	if (in_regs) {
#pragma unroll
		for (int k = 0; k < BSR_SLICE_REGS; k++) {
			const int i = sl.beg + tid + 256 * k;
			e[k] = i < sl.end ? load_elem_m(elems_in, (size_t)i, compact) : BinElem{0u, 0u, 0u};
		}
#pragma unroll
		for (int k = 0; k < BSR_SLICE_REGS; k++)
			if (sl.beg + tid + 256 * k < sl.end) atomicAdd(&s_hist[(e[k].x >> BSR_RADIX_BITS) & (BSR_RADIX_BINS - 1)], 1u);
	} else {
		slice_histogram(elems_in, sl, s_hist, compact);
	}
	__syncthreads();
	{
		const uint32_t c = s_hist[tid];
		const uint32_t t = ((uint32_t)tid << BSR_RADIX_BITS) | (uint32_t)d1;
		uint32_t off = 0;
		if (c != 0u && t < (uint32_t)T) off = tile_range[t].x + atomicAdd(&tile_cursor[t], c);
		s_off[tid] = off;
	}
	__syncthreads();
	if (in_regs) {
#pragma unroll
		for (int k = 0; k < BSR_SLICE_REGS; k++)
			if (sl.beg + tid + 256 * k < sl.end) {
				const uint32_t pos = atomicAdd(&s_off[(e[k].x >> BSR_RADIX_BITS) & (BSR_RADIX_BINS - 1)], 1u);   // LDS
				store_elem_m(elems_out, pos, e[k], compact);
			}
		return;
	}
	for (int i = sl.beg + tid; i < sl.end; i += 1024) {
		BinElem f[4];
#pragma unroll
		for (int k = 0; k < 4; k++)
			if (i + 256 * k < sl.end) f[k] = load_elem_m(elems_in, (size_t)(i + 256 * k), compact);
#pragma unroll
		for (int k = 0; k < 4; k++)
			if (i + 256 * k < sl.end) {
				const uint32_t pos = atomicAdd(&s_off[(f[k].x >> BSR_RADIX_BITS) & (BSR_RADIX_BINS - 1)], 1u);   // LDS
				store_elem_m(elems_out, pos, f[k], compact);
			}
	}
}

// ---- tile ranges: tile_start[t] = first sorted position whose tile id is >= t ----
// Tiles holding more than BSR_SORT_SMALL instances are also appended (one atomic per wave, order
// irrelevant) to the work list of their size class (see below); flags[1], [4], [5] count them.
#define BSR_SORT_SMALL BSR_SORT_SMALL_N
// First index in [0, n) whose tile id is >= t, found by the 16 lanes of a DPP row together: every round the lanes
// probe 16 evenly spaced positions of the remaining range (one dependent L2 load per round, 17-fold narrowing:
// 6 rounds for 3 M elements instead of the 22 of a binary search -- the kernel is pure load latency).
__device__ __forceinline__ int first_not_below_row16(const BinElem* __restrict__ elems_sorted, int n, uint32_t t,
                                                     int lane)
{
	const int j = lane & 15, sh = lane & 48;
	int lo = 0, hi = n;
	while (hi > lo) {   // uniform over the row; rows of one wave may need different round counts
		const int width = hi - lo;
		if (width <= 16) {
			const bool below = j < width && elems_sorted[lo + j].x < t;
			lo += __popc((uint32_t)(wave_ballot(below) >> sh) & 0xffffu);
			break;
		}
		const int p = lo + (int)(((uint64_t)width * (uint32_t)(j + 1)) / 17u);   // lo < p < hi, ascending in j
		const bool below = elems_sorted[p].x < t;
		const int c = __popc((uint32_t)(wave_ballot(below) >> sh) & 0xffffu);   // probes 0 .. c-1 are below t
		const int new_lo = c > 0 ? lo + (int)(((uint64_t)width * (uint32_t)c) / 17u) + 1 : lo;
		const int new_hi = c < 16 ? lo + (int)(((uint64_t)width * (uint32_t)(c + 1)) / 17u) : hi;
		lo = new_lo;
		hi = new_hi;
	}
	return lo;
}

// One wave per three tiles: row k (16 lanes) finds the start of tile 3w + k, k = 0..3; rows 0..2 own their tile
// (range written, size class decided with the next row's start as the end), row 3 only delivers the end of tile
// 3w + 2.
__global__ void __launch_bounds__(256) k_tile_ranges(int T, const int* __restrict__ n_ptr, int capacity,
                                                     const BinElem* __restrict__ elems_sorted,
                                                     uint2* __restrict__ tile_range,
                                                     uint32_t* __restrict__ big_tiles, int* __restrict__ flags)
{
	int n = *n_ptr;
	if (n > capacity) return;   // overflow: the stage is re-run
	if (n < 0) n = 0;
	const int lane = threadIdx.x & 63, row = lane >> 4;
	const int wave = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
	const int t = wave * 3 + row;
	bool big = false;
	int cnt_t = 0;
	{
		const int lo = (t <= T) ? first_not_below_row16(elems_sorted, n, (uint32_t)t, lane) : n;
		const int hi = __shfl_down(lo, 16, 64);   // the next row's start
		const bool owner = row < 3 && (lane & 15) == 0;
		if (owner && t < T) {
			tile_range[t] = make_uint2((uint32_t)lo, (uint32_t)hi);
			cnt_t = hi - lo;
			big = cnt_t > BSR_SORT_SMALL;
		}
	}
	// one work list per size class of the wide sort kernels (their grids are bounded per class):
	// class 0: (1024, 4096] -> big_tiles[0..T), count flags[1]; class 1: (4096, 8192] -> [T..2T), flags[4];
	// class 2: > 8192 -> [2T..3T), flags[5]
	int cls = -1;
	if (big) cls = cnt_t > 8192 ? 2 : (cnt_t > 4096 ? 1 : 0);
#pragma unroll
	for (int k = 0; k < 3; k++) {
		const uint64_t b = __ballot(cls == k);
		if (b == 0) continue;
		int base = 0;
		if (lane == 0) base = atomicAdd(&flags[k == 0 ? 1 : 3 + k], __popcll(b));
		base = __shfl(base, 0);
		if (cls == k) big_tiles[(size_t)k * T + base + __popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)t;
	}
}

// ---- per-tile bitonic sort of 64-bit keys ----
// Order = (tile, depth bits, Gaussian id) = the reference's stable radix-sort order (rasterizer_impl.cu:304-309): the
// tile part is done by the radix passes, here every tile's segment is sorted on the 64-bit key (depth bits, id).
// The network is the all-ascending form of bitonic sort (first step of every merge compares mirrored partners), so
// keys beyond n behave as +infinity pads.  Segments of up to 4096 keys are sorted in LDS by the round-based network
// below; longer ones (rare: a tile overlapped by > 4096 splats) run hybrid: every 4096-key chunk is sorted in LDS,
// then each merge does only its steps with partner distance >= 4096 in global memory (the plain steps right here)
// and finishes chunk by chunk in LDS.
#define BSR_PAD_KEY 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ void cx(uint64_t& a, uint64_t& b)
{
	const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
	a = lo;
	b = hi;
}
__device__ __forceinline__ void cmp_exchange(uint64_t* k, int lo, int hi)
{
	const uint64_t a = k[lo], b = k[hi];
	if (a > b) { k[lo] = b; k[hi] = a; }
}
// global-memory steps of the hybrid (a compare-exchange whose upper index is >= n is the no-op a pad needs)
template <int NT>
__device__ __forceinline__ void merge_mirror_step(uint64_t* k, int n, int n2, int size, int tid)
{
	const int half = size >> 1, sh = __builtin_ctz(half);
	for (int i = tid; i < (n2 >> 1); i += NT) {
		const int blk = i >> sh, off = i & (half - 1);
		const int hi = blk * size + size - 1 - off;
		if (hi < n) cmp_exchange(k, blk * size + off, hi);
	}
}
template <int NT>
__device__ __forceinline__ void merge_stride_step(uint64_t* k, int n, int n2, int stride, int tid)
{
	for (int i = tid; i < (n2 >> 1); i += NT) {
		const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
		const int hi = lo | stride;
		if (hi < n) cmp_exchange(k, lo, hi);
	}
}

__device__ __forceinline__ uint64_t elem_key(const BinElem e) { return ((uint64_t)e.z << 32) | (uint64_t)e.y; }

// ---- round-based network: 2^M keys per thread, M network steps per LDS round trip --------------------------------
// A thread holds K = 2^M keys of a round in registers: the M index bits a round's steps act on enumerate the thread's
// keys, every other bit comes from the thread id, so M steps run between one read and one write of the keys (8 keys:
// 3 steps).  With n2 / K <= 64 (two trips per round up to 128) a whole segment belongs to ONE wave and needs no
// workgroup barrier at all: the small class sorts four tiles per workgroup, one per wave.  (Its predecessor ran two
// steps per barrier with 4 keys per thread, half of its 256 threads idle on a 512-key tile: 65 % of its wave-cycles
// were barrier and LDS-latency waits.)
// A merge of runs into runs of `size` = first round: the mirrored step + strides size/4 .. size/2^M (the thread's
// key set {i0 ^ a (size - 1) ^ sum c_b t_b} is closed under all of them), then rounds of up to M plain strides down
// to 1.  Keys in the mirrored half are labelled with complemented stride bits so that every plain step orders
// "bit clear below bit set" in both halves.
// LDS bank swizzle.  The keys are 8 bytes: a ds_read_b64 is served in two groups of 32 lanes, conflict-free when the
// 32 key slots differ mod 32; a ds_write_b64 in four groups of 16 lanes, slots mod 16.  With keys at their natural
// index the short strides are 2- to 4-way conflicts on every access (PMC on the predecessor: SQ_LDS_BANK_CONFLICT =
// 61 % of its LDS cycles).  Key i lives in slot i ^ ((i >> M) & 31): a bijection of [0, n2) for every power of two n2
// (bits are only folded downwards), found by enumerating XOR-linear maps against the access pattern of every round
// (M zero bits inserted into the thread index at any position), the load and the read-out: all conflict-free.  It is
// linear over XOR, so a thread swizzles ONE index per round and reaches its other keys by XOR with wave-uniform
// constants.
template <int M> __device__ __forceinline__ constexpr int swz_m(int i) { return i ^ ((i >> M) & 31); }

// Compare-exchange flavours.  F64: the keys of a segment whose depth bits all lie in [0x00100000, 0x7ff00000) are
// positive, normal, finite doubles when read as binary64, and for those the unsigned order of the bit patterns IS
// the numeric order: v_min_f64 / v_max_f64 return one operand unchanged each, two instructions instead of a 64-bit
// compare and four selects (selects and compares issue at 4.25 cycles on gfx950, the sort is bound by exactly these).
// The pad is +infinity (above every such key).  Segments holding any other depth pattern (NaN payloads, denormal or
// non-positive depths: the reference orders them by raw bits too) take the integer flavour.
#define BSR_PAD_F64 0x7FF0000000000000ull
template <bool F64>
__device__ __forceinline__ void cxt(uint64_t& a, uint64_t& b)
{
	if constexpr (F64) {
		double lo, hi;
		const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b);
		asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
		asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
		a = (uint64_t)__double_as_longlong(lo);
		b = (uint64_t)__double_as_longlong(hi);
	} else {
		cx(a, b);
	}
}
__device__ __forceinline__ bool key_is_plain_double(uint64_t k)
{
	const uint32_t h = (uint32_t)(k >> 32);
	return h >= 0x00100000u && h < 0x7ff00000u;
}

template <int M, int BIT, bool F64>
__device__ __forceinline__ void reg_step(uint64_t (&e)[1 << M])
{
#pragma unroll
	for (int c = 0; c < (1 << M); c++)
		if (!(c & (1 << BIT))) cxt<F64>(e[c], e[c | (1 << BIT)]);
}
// plain steps on local bits NS-1 .. 0
template <int M, int NS, bool F64>
__device__ __forceinline__ void reg_steps(uint64_t (&e)[1 << M])
{
	if constexpr (NS > 0) {
		reg_step<M, NS - 1, F64>(e);
		reg_steps<M, NS - 1, F64>(e);
	}
}
// mirrored step: local index (a, c), a = top bit: (0, c) <-> (1, ~c)
template <int M, bool F64>
__device__ __forceinline__ void reg_mirror(uint64_t (&e)[1 << M])
{
	constexpr int H = 1 << (M - 1);
#pragma unroll
	for (int c = 0; c < H; c++) cxt<F64>(e[c], e[H + (H - 1 - c)]);
}
// the K keys of a thread, ascending, entirely in registers (bitonic: sizes 2 .. K)
template <int M, bool F64>
__device__ __forceinline__ void reg_sort(uint64_t (&e)[1 << M])
{
#pragma unroll
	for (int sbit = 1; sbit <= M; sbit++) {
#pragma unroll
		for (int c = 0; c < (1 << M); c++)
			if (!(c & (1 << (sbit - 1)))) {
				const int partner = c ^ ((1 << sbit) - 1);
				cxt<F64>(e[c], e[partner]);
			}
#pragma unroll
		for (int b = sbit - 2; b >= 0; b--)
#pragma unroll
			for (int c = 0; c < (1 << M); c++)
				if (!(c & (1 << b))) cxt<F64>(e[c], e[c | (1 << b)]);
	}
}

// byte offset of local key L from the thread's first slot: XOR of the deltas of L's set bits (all wave-uniform)
template <int M>
__device__ __forceinline__ int local_delta(const int (&d)[M], int L)
{
	int x = 0;
#pragma unroll
	for (int b = 0; b < M; b++)
		if (L & (1 << b)) x ^= d[b];
	return x;
}
template <int M>
__device__ __forceinline__ void round_load(const char* lds, int p0, const int (&d)[M], uint64_t (&e)[1 << M])
{
#pragma unroll
	for (int L = 0; L < (1 << M); L++) e[L] = *reinterpret_cast<const uint64_t*>(lds + (p0 ^ local_delta<M>(d, L)));
}
template <int M>
__device__ __forceinline__ void round_store(char* lds, int p0, const int (&d)[M], const uint64_t (&e)[1 << M])
{
#pragma unroll
	for (int L = 0; L < (1 << M); L++) *reinterpret_cast<uint64_t*>(lds + (p0 ^ local_delta<M>(d, L))) = e[L];
}

template <bool BLOCK>
__device__ __forceinline__ void round_sync()
{
	if (BLOCK) {
		__syncthreads();
	} else {   // one wave owns the segment: its LDS operations execute in order; only the compiler must not reorder
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
}

// The plain strides 2^(rem-1) .. 1 of a merge, M per round, over n2 keys (slots swz_m<M>); t = thread index among
// the NT threads that share the segment.
template <int NT, int M, bool BLOCK, bool F64>
__device__ __forceinline__ void lds_stride_rounds(uint64_t* keys, int n2, int rem, int t)
{
	char* const lds = reinterpret_cast<char*>(keys);
	while (rem > 0) {
		const int ns = min(M, rem);
		const int lo = max(rem - M, 0);   // local bit b <-> index bit lo + b; a last, short round steps on bits ns-1 .. 0 only
		int d[M];
#pragma unroll
		for (int b = 0; b < M; b++) d[b] = swz_m<M>(1 << (lo + b)) << 3;
		round_sync<BLOCK>();
		for (int i = t; i < (n2 >> M); i += NT) {
			const int i0 = ((i >> lo) << (lo + M)) | (i & ((1 << lo) - 1));
			const int p0 = swz_m<M>(i0) << 3;
			uint64_t e[1 << M];
			round_load<M>(lds, p0, d, e);
			if (ns == M) reg_steps<M, M, F64>(e);
			else if (M > 3 && ns == 3) reg_steps<M, (M > 3 ? 3 : 1), F64>(e);
			else if (ns == 2) reg_steps<M, 2, F64>(e);
			else reg_steps<M, 1, F64>(e);
			round_store<M>(lds, p0, d, e);
		}
		rem -= ns;
	}
}

// Ascending sort of n2 (a power of two >= 2^M) keys in LDS whose aligned runs of 2^M are sorted already; pads
// (BSR_PAD_KEY) are ordinary keys.  Ends with a sync.  (Skipping the work items of blocks that hold pads only -- 20 % of
// the items of a 1200-key segment in 2048 slots -- was measured in round 6: no change in either sort kernel.)
template <int NT, int M, bool BLOCK, bool F64>
__device__ __forceinline__ void lds_sort_rounds(uint64_t* keys, int n2, int t)
{
	static_assert(M == 3 || M == 4, "8 or 16 keys per thread");
	char* const lds = reinterpret_cast<char*>(keys);
	for (int size = 2 << M; size <= n2; size <<= 1) {
		const int k = __builtin_ctz(size), lo = k - M, tlow = 1 << lo;
		// first round: mirror + strides size/4 .. size/2^M.  Local bits 0 .. M-2 <-> strides tlow << b; the top local
		// bit selects the mirrored half, whose keys carry complemented stride bits: its delta is (size - 1) ^ all strides
		int d[M], low_all = 0;
#pragma unroll
		for (int b = 0; b < M - 1; b++) {
			d[b] = swz_m<M>(tlow << b) << 3;
			low_all ^= tlow << b;
		}
		d[M - 1] = swz_m<M>((size - 1) ^ low_all) << 3;
		round_sync<BLOCK>();
		for (int i = t; i < (n2 >> M); i += NT) {
			const int i0 = ((i >> lo) << k) | (i & (tlow - 1));
			const int p0 = swz_m<M>(i0) << 3;
			uint64_t e[1 << M];
			round_load<M>(lds, p0, d, e);
			reg_mirror<M, F64>(e);
			reg_steps<M, M - 1, F64>(e);
			round_store<M>(lds, p0, d, e);
		}
		lds_stride_rounds<NT, M, BLOCK, F64>(keys, n2, k - M, t);
	}
	round_sync<BLOCK>();
}

// One segment of n <= n2 keys, sorted by the NT threads (thread index t) that share `keys` (n2 slots).  Runs of 2^M
// are sorted in registers on the way in (integer compare-exchange: the flavour of the merges is only known once every
// key has been seen) and the pads up to n2 are stored with them; returns this thread's vote on "every key I loaded is
// a positive, normal, finite binary64".
// Where a segment's unsorted keys come from: the binning elements (8- or 12-byte form), or plain 64-bit keys
// (k_bucket_sort stages its long tiles that way).  operator()(i) = key at global position i.
struct ElemKeys {
	const BinElem* __restrict__ elems;
	int compact;
	__device__ __forceinline__ uint64_t operator()(size_t i) const { return elem_key_m(elems, i, compact); }
};
struct RawKeys {
	const uint64_t* keys;   // (no __restrict__: k_bucket_sort writes the scratch it then sorts from)
	__device__ __forceinline__ uint64_t operator()(size_t i) const { return keys[i]; }
};
template <int NT, int M, typename Src>
__device__ __forceinline__ bool load_sorted_runs(uint64_t* keys, int n2, uint32_t start, int n, int t, const Src src)
{
	constexpr int K = 1 << M;
	bool plain = true;
	for (int i = t * K; i < n2; i += NT * K) {
		uint64_t e[K];
#pragma unroll
		for (int j = 0; j < K; j++) {
			e[j] = i + j < n ? src((size_t)start + (size_t)(i + j)) : BSR_PAD_KEY;
			plain = plain && (i + j >= n || key_is_plain_double(e[j]));
		}
		reg_sort<M, false>(e);
		const int p0 = swz_m<M>(i);   // i is a multiple of K: i + j == i ^ j
#pragma unroll
		for (int j = 0; j < K; j++) keys[p0 ^ swz_m<M>(j)] = e[j];
	}
	return plain;
}
template <int NT, int M, bool BLOCK, bool F64>
__device__ __forceinline__ void merge_loaded_runs(uint64_t* keys, int n2, uint32_t start, int n, int t,
                                                  uint32_t* __restrict__ point_list)
{
	if (F64) {   // the pads become +infinity (they sit at the ends of their runs either way)
		round_sync<BLOCK>();
		for (int i = n + t; i < n2; i += NT) keys[swz_m<M>(i)] = BSR_PAD_F64;
	}
	lds_sort_rounds<NT, M, BLOCK, F64>(keys, n2, t);
	for (int i = t; i < n; i += NT) point_list[start + i] = (uint32_t)keys[swz_m<M>(i)];
}

// ---- bucket-and-rank sort of one segment (round 6; the network above stays as the fall-back) -----------------------
// A segment's keys are (depth bits, id), and the depth bits of the splats over one tile spread over their range: instead
// of n log^2 n compare-exchanges the keys are dealt into NB = 2^NBLOG buckets by a MONOTONE map of the depth
// (common.h: rank_sort_bucket),
//     b = min(int((z - z_min) * (NB - 0.5) / (z_max - z_min)), NB - 1),
// (histogram with LDS atomics, one scan, one scatter: the keys then lie bucket by bucket, in arrival order inside a
// bucket), and a key's final place is its bucket's first position + the number of smaller keys in its bucket, counted
// against the bucket's members (keys are unique within a tile: ids are).  A key goes through LDS once (8-byte write,
// 8-byte read) plus ~2 reads per fellow member; the ids are written to point_list straight from the count.  The result is
// THE ascending order of the keys -- the same bits as the network's -- for every input; what depends on the input is
// only the price: a segment with a bucket of more than BSR_RANK_CAP keys (depths piled on one value), or with a depth
// word that is not a positive finite float (the reference orders those by raw bits too), is left untouched and the
// caller sorts it with the network.  Counters are 16 bits wide, two per dword (counts and offsets <= 4096): the
// low one cannot carry into the high one.
//   NT threads (t = index) share the segment; thread t holds keys i = t + NT q, q < KPT, in registers (valid: i < n) --
//   `out` (n slots of LDS) may therefore be the very area the keys were read from; cnt: NB / 2 dwords of LDS;
//   s_red (BLOCK only): 3 * NT / 64 dwords.  Ends without a sync: the caller syncs before `out` / `cnt` are reused.
#ifndef BSR_RANK_CAP
#define BSR_RANK_CAP 32
#endif
// maximum over the 64 lanes on the vector ALU (the steps of wave_inclusive_sum_dpp with max for +: lanes without a source
// take 0, the identity of an unsigned max; lane 63 ends with the total).  (Six __shfl_xor steps are six dependent
// ds_bpermute round trips: ~700 cycles per reduction, three reductions per sorted segment.)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
	x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));   // row_shr:1
	x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));   // row_shr:2
	x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));   // row_shr:4
	x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));   // row_shr:8
	x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
	x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
	return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
template <int NT, int KPT, int NBLOG, bool BLOCK>
__device__ __forceinline__ bool rank_sort(const uint64_t (&e)[KPT], int n, int t, uint64_t* out, uint32_t* cnt,
                                          uint32_t* s_red, uint32_t start, uint32_t* __restrict__ point_list)
{
	constexpr int NB = 1 << NBLOG, NDW = NB / 2, DPT = NDW / NT, NWV = NT / 64;
	static_assert(NDW % NT == 0 && DPT >= 1 && DPT <= 8, "the scan takes up to 8 counter dwords per thread");
	static_assert(KPT % 4 == 0, "the read-out takes four keys per thread and trip");
	const int lane = t & 63, wave = t >> 6;
	// ---- the range of the depth words
	uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
	for (int q = 0; q < KPT; q++)
		if (t + NT * q < n) {
			const uint32_t h = (uint32_t)(e[q] >> 32);
			lo = min(lo, h);
			hi = max(hi, h);
		}
	lo = ~wave_max_u32(~lo);
	hi = wave_max_u32(hi);
	if (BLOCK) {
		if (lane == 0) {
			s_red[wave] = lo;
			s_red[NWV + wave] = hi;
		}
		__syncthreads();
#pragma unroll
		for (int w = 0; w < NWV; w++) {
			lo = min(lo, s_red[w]);
			hi = max(hi, s_red[NWV + w]);
		}
	}
	if (lo == 0u || hi >= 0x7f800000u) return false;   // (uniform) not all positive finite floats: the network's integer flavour
	const float zlo = __uint_as_float(lo), scale = rank_sort_scale(zlo, __uint_as_float(hi), NB);
	// ---- histogram; the returning atomic also tells a key how many keys were in its bucket before it -- its place in the
	// bucket's run (one byte each: a count past 255 wraps, and such a segment is declined below)
#pragma unroll
	for (int w = 0; w < DPT; w++) cnt[t * DPT + w] = 0u;
	round_sync<BLOCK>();
	uint32_t arrival[KPT / 4];
#pragma unroll
	for (int q = 0; q < KPT / 4; q++) arrival[q] = 0u;
#pragma unroll
	for (int q = 0; q < KPT; q++)
		if (t + NT * q < n) {
			const uint32_t b = rank_sort_bucket(__uint_as_float((uint32_t)(e[q] >> 32)), zlo, scale, NB), sh = (b & 1u) << 4;
			const uint32_t before = (atomicAdd(&cnt[b >> 1], 1u << sh) >> sh) & 0xffu;   // LDS
			arrival[q >> 2] |= before << ((q & 3) << 3);
		}
	round_sync<BLOCK>();
	// ---- exclusive scan of the NB counts: thread t owns buckets 2 DPT t .. 2 DPT (t + 1) - 1
	uint32_t c[2 * DPT], total = 0u, cmax = 0u;
#pragma unroll
	for (int w = 0; w < DPT; w++) {
		const uint32_t v = cnt[t * DPT + w];
		c[2 * w] = v & 0xffffu;
		c[2 * w + 1] = v >> 16;
		total += c[2 * w] + c[2 * w + 1];
		cmax = max(cmax, max(c[2 * w], c[2 * w + 1]));
	}
	const uint32_t incl = wave_inclusive_sum_dpp(total);
	cmax = wave_max_u32(cmax);
	uint32_t run = incl - total;
	if (BLOCK) {
		if (lane == 63) s_red[2 * NWV + wave] = incl;
		__syncthreads();   // (also: every thread has read lo / hi above)
		if (lane == 0) s_red[wave] = cmax;
		for (int w = 0; w < wave; w++) run += s_red[2 * NWV + w];
		__syncthreads();
#pragma unroll
		for (int w = 0; w < NWV; w++) cmax = max(cmax, s_red[w]);
	}
	if (cmax > (uint32_t)BSR_RANK_CAP) return false;   // (uniform over the NT threads; nothing but cnt was written)
#pragma unroll
	for (int w = 0; w < DPT; w++) {
		const uint32_t o0 = run, o1 = run + c[2 * w];
		run = o1 + c[2 * w + 1];
		cnt[t * DPT + w] = o0 | (o1 << 16);
	}
	round_sync<BLOCK>();
	// ---- scatter: the keys bucket by bucket (every thread holds its keys in registers: `out` may be their old place)
	const uint16_t* const first = reinterpret_cast<const uint16_t*>(cnt);   // first position of every bucket
#pragma unroll
	for (int q = 0; q < KPT; q++)
		if (t + NT * q < n) {
			const uint32_t b = rank_sort_bucket(__uint_as_float((uint32_t)(e[q] >> 32)), zlo, scale, NB);
			out[(uint32_t)first[b] + ((arrival[q >> 2] >> ((q & 3) << 3)) & 0xffu)] = e[q];
		}
	round_sync<BLOCK>();
	// ---- a key's place = first position of its bucket + the number of smaller keys in the bucket
	constexpr int CHQ = 4;   // keys per thread and trip (all KPT at once: 5 KPT live registers)
#pragma unroll 1
	for (int q0 = 0; q0 < KPT && NT * q0 < n; q0 += CHQ) {
		uint64_t k[CHQ];
		uint32_t beg[CHQ], len[CHQ], rank[CHQ];
#pragma unroll
		for (int q = 0; q < CHQ; q++) {
			const int p = t + NT * (q0 + q);
			k[q] = 0ull;
			beg[q] = len[q] = rank[q] = 0u;
			if (p < n) {
				k[q] = out[p];
				const uint32_t b = rank_sort_bucket(__uint_as_float((uint32_t)(k[q] >> 32)), zlo, scale, NB);
				beg[q] = (uint32_t)first[b];
				const uint32_t end = b + 1u < (uint32_t)NB ? (uint32_t)first[b + 1u] : (uint32_t)n;
				len[q] = min(end - beg[q], (uint32_t)BSR_RANK_CAP);   // (<= the cap by the vote above: the clamp only bounds the
				                                                      // loop below whatever LDS holds)
			}
		}
		// JU members per key and trip: 4 JU independent LDS reads in flight (one member per trip left the loop at one
		// LDS round trip per member of the fullest bucket); the wave stops when its longest bucket is through
		uint32_t longest = max(max(len[0], len[1]), max(len[2], len[3]));
#ifndef BSR_RANK_JU_BLOCK
#define BSR_RANK_JU_BLOCK 2
#endif
		constexpr int JU = BLOCK ? BSR_RANK_JU_BLOCK : 4;   // (the workgroup-owned flavour runs in k_sort_tiles_wide's 80 VGPRs)
		for (uint32_t j0 = 0; wave_ballot(j0 < longest) != 0ull; j0 += JU) {
			uint64_t mem[CHQ][JU];
#pragma unroll
			for (int q = 0; q < CHQ; q++)
#pragma unroll
				for (int u = 0; u < JU; u++) mem[q][u] = out[min(beg[q] + j0 + u, (uint32_t)(n - 1))];   // (unconditional reads)
#pragma unroll
			for (int q = 0; q < CHQ; q++)
#pragma unroll
				for (int u = 0; u < JU; u++) rank[q] += (j0 + u < len[q] && mem[q][u] < k[q]) ? 1u : 0u;
		}
#pragma unroll
		for (int q = 0; q < CHQ; q++)
			if (t + NT * (q0 + q) < n) point_list[start + beg[q] + rank[q]] = (uint32_t)k[q];
	}
	return true;
}

// the same with the keys read from global memory (thread t: positions start + t + NT q, coalesced)
template <int NT, int KPT, int NBLOG, bool BLOCK, typename Src>
__device__ __forceinline__ bool rank_sort_from(const Src src, int n, int t, uint64_t* out, uint32_t* cnt, uint32_t* s_red,
                                               uint32_t start, uint32_t* __restrict__ point_list)
{
	uint64_t e[KPT];
#pragma unroll
	for (int q = 0; q < KPT; q++) e[q] = t + NT * q < n ? src((size_t)start + (size_t)(t + NT * q)) : 0ull;
	return rank_sort<NT, KPT, NBLOG, BLOCK>(e, n, t, out, cnt, s_red, start, point_list);
}

// Wave-owned segment: load, pick the compare-exchange flavour, sort.  No workgroup barrier anywhere.
template <int M>
__device__ __forceinline__ void sort_segment_wave(uint64_t* keys, int n2, uint32_t start, int n, int lane,
                                                  const BinElem* __restrict__ elems, uint32_t* __restrict__ point_list,
                                                  bool force_int, int compact)
{
	const bool plain = load_sorted_runs<64, M>(keys, n2, start, n, lane, ElemKeys{elems, compact}) && !force_int;
	if (wave_ballot(!plain) == 0ull)
		merge_loaded_runs<64, M, false, true>(keys, n2, start, n, lane, point_list);
	else
		merge_loaded_runs<64, M, false, false>(keys, n2, start, n, lane, point_list);
}

// Workgroup-owned segment (the wide classes): the same, with workgroup barriers and a workgroup vote.
template <int NT, int M, typename Src>
__device__ __forceinline__ void sort_segment_block(uint64_t* keys, int n2, uint32_t start, int n, int tid, const Src src,
                                                   uint32_t* __restrict__ point_list, bool force_int)
{
	const bool plain = load_sorted_runs<NT, M>(keys, n2, start, n, tid, src) && !force_int;
	if (__syncthreads_and(plain))
		merge_loaded_runs<NT, M, true, true>(keys, n2, start, n, tid, point_list);
	else
		merge_loaded_runs<NT, M, true, false>(keys, n2, start, n, tid, point_list);
}

// Tiny class (n <= 64: every tile of a sparse camera-sweep view): one wave per tile, one key per lane, no LDS and no
// synchronisation at all -- a key's place is the number of smaller keys ((depth bits, id) pairs are unique within a
// tile), counted against the wave's keys broadcast one by one from SGPRs.  27 keys: ~110 instructions, against a
// merge network that keeps 4 of 64 lanes busy and a 32-KB LDS footprint that caps the small class at 20 waves per CU.
__global__ void __launch_bounds__(256) k_sort_tiles_tiny(int T, const int* __restrict__ n_ptr, int capacity,
                                                         const uint2* __restrict__ tile_range,
                                                         const BinElem* __restrict__ elems,
                                                         uint32_t* __restrict__ point_list, int compact)
{
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (tile >= T) return;
	const int n_instances = *n_ptr;
	const uint2 range = tile_range[tile];
	const uint32_t start = range.x;
	const int n = (int)(range.y - range.x);
	if (n_instances > capacity || n > 64 || n <= 0) return;   // (scratch too small: stage is re-run) / another class / empty
	uint64_t key = ~0ull;
	if (lane < n) key = elem_key_m(elems, (size_t)start + (size_t)lane, compact);
	const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
	uint32_t rank = 0;
	for (int j = 0; j < n; j++) {
		const uint64_t kj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, j) << 32) |
		                    (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)lo, j);
		rank += kj < key ? 1u : 0u;
	}
	if (lane < n) point_list[start + rank] = lo;
}

// Small class (min_n < n <= BSR_SORT_SMALL): one WAVE per tile, four tiles per workgroup, no workgroup barrier; 8 keys per
// lane and round (16 in two trips beyond 512 keys).
__global__ void __launch_bounds__(256) k_sort_tiles_small(int T, const int* __restrict__ n_ptr, int capacity,
                                                          const uint2* __restrict__ tile_range,
                                                          const BinElem* __restrict__ elems,
                                                          uint32_t* __restrict__ point_list, int sort_mode, int min_n,
                                                          int compact)
{
	__shared__ uint64_t s_keys[4][BSR_SORT_SMALL];
	__shared__ uint32_t s_rank[4][512];   // rank_sort's counters: up to 1024 buckets per wave
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + wave;
	if (tile >= T || *n_ptr > capacity) return;   // (more instances than the scratch was sized for: stage is re-run)
	const uint2 range = tile_range[tile];
	const uint32_t start = range.x;
	const int n = (int)(range.y - range.x);
	if (n > BSR_SORT_SMALL || n <= min_n) return;   // on the big-tile list / sorted by k_sort_tiles_tiny (min_n = 64) or empty
	// bucket-and-rank sort first (sort_mode 0); a segment it declines -- depths piled on one value -- goes to the network
	if (sort_mode == 0 && n > 64) {
		if (n <= 512) {
			if (rank_sort_from<64, 8, 9, false>(ElemKeys{elems, compact}, n, lane, s_keys[wave], s_rank[wave], nullptr, start, point_list)) return;
		} else {
			if (rank_sort_from<64, 16, 10, false>(ElemKeys{elems, compact}, n, lane, s_keys[wave], s_rank[wave], nullptr, start, point_list)) return;
		}
		round_sync<false>();
	}
	int n2 = 8;
	while (n2 < n) n2 <<= 1;
	sort_segment_wave<3>(s_keys[wave], n2, start, n, lane, elems, point_list, (sort_mode & 1) != 0, compact);   // (> 512 keys: two runs per lane)
}

// One long segment, 1024 < n <= BSR_SORT_CHUNK keys, sorted in `s_keys` (BSR_SORT_CHUNK slots) by the NT threads of
// the workgroup and read out to point_list[start ..).  Ends with a barrier (the keys are read out before the caller
// loads the next segment).
#define BSR_SORT_CHUNK 4096
template <int NT, typename Src>
__device__ __forceinline__ void sort_long_tile_lds(uint64_t* s_keys, uint32_t* s_rank, uint32_t start, int n, int tid,
                                                   const Src src, uint32_t* __restrict__ point_list, int sort_mode)
{
	// bucket-and-rank sort first (sort_mode 0: 8 keys per thread = up to 8 NT keys, 4 NT buckets: s_rank holds 2 NT counter
	// dwords + the reduction words); declined segments go to the network
	if (sort_mode == 0) {
		constexpr int NBLOG = NT == 512 ? 11 : 10;
		static_assert(NT == 512 || NT == 256, "4096- or 2048-key segments");
		const bool done = rank_sort_from<NT, 8, NBLOG, true>(src, n, tid, s_keys, s_rank, s_rank + 2 * NT, start, point_list);
		__syncthreads();
		if (done) return;
	}
	int n2 = 1024;
	while (n2 < n) n2 <<= 1;
	sort_segment_block<NT, 3>(s_keys, n2, start, n, tid, src, point_list, (sort_mode & 1) != 0);
	__syncthreads();
}
// One segment of n > BSR_SORT_CHUNK keys, hybrid: every 4096-key chunk sorted in LDS into the global scratch k[0 .. n)
// (`src` may read that very scratch: a chunk is loaded completely before it is written back), the merge steps between
// chunks in global memory, the steps inside a chunk in LDS again.  Integer compare-exchange throughout (the global
// steps compare integers too).
template <int NT, typename Src>
__device__ __forceinline__ void sort_long_tile_hybrid(uint64_t* s_keys, uint64_t* k, uint32_t start, int n, int tid,
                                                      const Src src, uint32_t* __restrict__ point_list)
{
	constexpr int CH = BSR_SORT_CHUNK;
	int n2 = 1;
	while (n2 < n) n2 <<= 1;
	// runs of CH: every chunk sorted on its own in LDS
	for (int base = 0; base < n; base += CH) {
		const int m = min(CH, n - base);
		__syncthreads();
		load_sorted_runs<NT, 3>(s_keys, CH, start + (uint32_t)base, m, tid, src);
		lds_sort_rounds<NT, 3, true, false>(s_keys, CH, tid);
		for (int i = tid; i < m; i += NT) k[base + i] = s_keys[swz_m<3>(i)];
	}
	// merges of runs longer than CH: far partners in global memory, the rest per chunk in LDS
	for (int size = 2 * CH; size <= n2; size <<= 1) {
		__syncthreads();
		merge_mirror_step<NT>(k, n, n2, size, tid);
		for (int stride = size >> 2; stride >= CH; stride >>= 1) {
			__syncthreads();
			merge_stride_step<NT>(k, n, n2, stride, tid);
		}
		for (int base = 0; base < n; base += CH) {
			const int m = min(CH, n - base);
			__syncthreads();
			for (int i = tid; i < CH; i += NT) s_keys[swz_m<3>(i)] = i < m ? k[base + i] : BSR_PAD_KEY;
			lds_stride_rounds<NT, 3, true, false>(s_keys, CH, 12, tid);   // strides CH/2 .. 1
			__syncthreads();
			for (int i = tid; i < m; i += NT) k[base + i] = s_keys[swz_m<3>(i)];
		}
	}
	__syncthreads();
	for (int i = tid; i < n; i += NT) point_list[start + i] = (uint32_t)k[i];
	__syncthreads();
}

// Wide classes, ONE launch (a frame without long lists -- C3 -- pays one near-empty launch instead of two; until round 5
// the two upper classes had a 1024-thread, 64-KB kernel of their own): 512 threads, 4096 keys = 32 KB of LDS.
// Workgroups [0, g1) stride over the (1024, 4096] list (big_tiles[0..T), count flags[1]) and sort each segment in LDS;
// workgroups [g1, g1 + gw) stride over the two longer lists (big_tiles[T..2T), flags[4]; [2T..3T), flags[5]) with the
// hybrid: every 4096-key chunk sorted in LDS, the merge steps between chunks in global scratch (`keys` = the free
// ping-pong buffer viewed as u64), the steps inside a chunk in LDS again.  Bounded grids: n instances fill at most
// n / 1025 (n / 4097) such tiles, capped -- the workgroups stride.
#define BSR_SORT_NT 512
// (64 VGPRs: with 33 KB of LDS a CU holds four workgroups = 8 waves per SIMD; the hybrid path alone would take 70 and
// cost the common (1024, 4096] class its fourth workgroup: C5's tile sort 0.184 -> 0.206 ms)
#ifndef BSR_WIDE_WAVES
#define BSR_WIDE_WAVES 6
#endif
// The lower half of the first wide class, (1024, 2048] keys, in a launch of its own (round 6): 256 threads, 16 KB of keys +
// 2 KB of counters.  A segment's sort is short (~3 us); what a workgroup of k_sort_tiles_wide spends per segment is
// mostly the chain of dependent loads ahead of it (list entry -> range -> keys) and the drain of its stores behind it.
// Here the workgroups are few enough to be resident all at once and stride over the work list (big_tiles[0..flags[1]))
// with the loads of the NEXT segments in flight under the sort of the current one: the range two entries ahead, the keys
// (8 per thread, in registers) one entry ahead.  k_sort_tiles_wide skips what is sorted here.
#define BSR_SORT_MID 2048
#ifndef BSR_SORT_MID_WGS
#define BSR_SORT_MID_WGS 1280   // five workgroups per CU
#endif
#ifndef BSR_MID_NBLOG
#define BSR_MID_NBLOG 11   // 2048 buckets for up to 2048 keys (1024: +2 us on the dense leg)
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) k_sort_tiles_mid(int g1, const int* __restrict__ n_ptr, int capacity,
                                                                 const uint2* __restrict__ tile_range,
                                                                 const uint32_t* __restrict__ big_tiles,
                                                                 const int* __restrict__ flags,
                                                                 const BinElem* __restrict__ elems,
                                                                 uint32_t* __restrict__ point_list, int sort_mode, int compact)
{
	__shared__ uint64_t s_keys[BSR_SORT_MID];
	__shared__ uint32_t s_rank[(1 << BSR_MID_NBLOG) / 2 + 16];   // rank_sort's counters + its reduction words
	const int tid = threadIdx.x;
	if (*n_ptr > capacity) return;
	const ElemKeys src{elems, compact};
	const int count = flags[1];
	// the segment of list entry b if it belongs to this class, else an empty one
	auto segment = [&](int b) {
		if (b >= count) return make_uint2(0u, 0u);
		const uint2 r = tile_range[big_tiles[b]];
		const int n = (int)(r.y - r.x);
		return (n > BSR_SORT_SMALL && n <= BSR_SORT_MID) ? r : make_uint2(0u, 0u);
	};
	auto load_keys = [&](const uint2 r, uint64_t (&e)[8]) {
		const int n = (int)(r.y - r.x);
#pragma unroll
		for (int q = 0; q < 8; q++) e[q] = tid + 256 * q < n ? src((size_t)r.x + (size_t)(tid + 256 * q)) : 0ull;
	};
	uint2 cur = segment((int)blockIdx.x), nxt = segment((int)blockIdx.x + g1);
	uint64_t e[8];
	load_keys(cur, e);
	for (int b = blockIdx.x; b < count; b += g1) {
		const uint2 nxt2 = segment(b + 2 * g1);
		uint64_t en[8];
		load_keys(nxt, en);
		const int n = (int)(cur.y - cur.x);
		if (n > 0) {   // (uniform over the workgroup)
			bool done = false;
			if (sort_mode == 0) {   // bucket-and-rank sort; a declined segment goes through the network (which loads it again)
				done = rank_sort<256, 8, BSR_MID_NBLOG, true>(e, n, tid, s_keys, s_rank, s_rank + (1 << BSR_MID_NBLOG) / 2, cur.x, point_list);
				__syncthreads();
#ifdef BSR_MID_TWICE   // (cost attribution: the sort a second time, same keys, same result)
				done = rank_sort<256, 8, BSR_MID_NBLOG, true>(e, n, tid, s_keys, s_rank, s_rank + (1 << BSR_MID_NBLOG) / 2, cur.x, point_list);
				__syncthreads();
#endif
			}
			if (!done) {
				sort_segment_block<256, 3>(s_keys, BSR_SORT_MID, cur.x, n, tid, src, point_list, (sort_mode & 1) != 0);
				__syncthreads();
			}
		}
		cur = nxt;
		nxt = nxt2;
#pragma unroll
		for (int q = 0; q < 8; q++) e[q] = en[q];
	}
}

__global__ void __launch_bounds__(BSR_SORT_NT) __attribute__((amdgpu_waves_per_eu(BSR_WIDE_WAVES, 8))) k_sort_tiles_wide(int T, int g1, const int* __restrict__ n_ptr, int capacity,
                                                                 const uint2* __restrict__ tile_range,
                                                                 const uint32_t* __restrict__ big_tiles,
                                                                 const int* __restrict__ flags,
                                                                 const BinElem* __restrict__ elems, uint64_t* keys,
                                                                 uint32_t* __restrict__ point_list, int sort_mode, int compact,
                                                                 int lds_min)   // segments of up to lds_min keys: another kernel's
{
	constexpr int NT = BSR_SORT_NT, CH = BSR_SORT_CHUNK;
	__shared__ uint64_t s_keys[CH];
	__shared__ uint32_t s_rank[1024 + 32];   // rank_sort's counters (2048 buckets) + its reduction words
	const int tid = threadIdx.x;
	if (*n_ptr > capacity) return;
	const ElemKeys src{elems, compact};
	if ((int)blockIdx.x < g1) {
		const int count = flags[1];
		for (int b = blockIdx.x; b < count; b += g1) {
			const uint32_t tile = big_tiles[b];
			const uint2 range = tile_range[tile];
			const uint32_t start = range.x;
			const int n = (int)(range.y - range.x);
			if (n <= lds_min || n > CH) continue;   // another class (uniform over the workgroup)
			sort_long_tile_lds<NT>(s_keys, s_rank, start, n, tid, src, point_list, sort_mode);
		}
		return;
	}
	const int gw = (int)gridDim.x - g1, count4 = flags[4], count8 = flags[5];
	for (int b = (int)blockIdx.x - g1; b < count4 + count8; b += gw) {
		const uint32_t tile = b < count4 ? big_tiles[(size_t)T + b] : big_tiles[2 * (size_t)T + (b - count4)];
		const uint2 range = tile_range[tile];
		const uint32_t start = range.x;
		const int n = (int)(range.y - range.x);
		if (n <= CH) continue;
		sort_long_tile_hybrid<NT>(s_keys, keys + start, start, n, tid, src, point_list);
	}
}

// ---- bucket-owned second pass + per-tile sort, ONE launch (frames of up to 8192 tiles with short lists) ----------------
// After pass 1 (k_emit_scatter) the instances of tile t all lie in bucket t & 255, a contiguous range whose bounds follow
// from the 256 digit totals alone.  Nothing downstream needs the tile segments in TILE order -- the tile walks and the
// backward take (start, end) per tile from tile_range -- so the segments of a bucket's tiles can simply be laid out
// inside the bucket's own range: a segment's position then depends on the counts of ITS bucket only, and the chain
// k_tile_count -> k_tile_starts (one workgroup, a global scan) -> k_tile_scatter -> k_sort_tiles_small ->
// k_sort_tiles_wide  (five launches, the elements written and read once more) collapses into one kernel without any
// communication between workgroups:
//   workgroup (bucket d, part j of k = 2^k_log2): owns the bucket's tiles whose high byte hi = j (mod k) -- at most
//   NW * TPW of them, tile L = hi / k in LDS area L -- and streams the WHOLE bucket once (8-byte elements; the k parts
//   of a bucket run on one XCD back to back: one HBM read, k - 1 L2 hits).  An element of one of its tiles takes its
//   slot in the tile's area from an LDS counter (= the tile's count in the end); of the others only those of EARLIER
//   parts are counted (one wave ballot per element, no LDS traffic), which is all the layout needs: the bucket's range
//   holds part 0's tiles, then part 1's, ..., inside a part in order of L -- part j begins behind the elements of the
//   parts before it, which every workgroup of the bucket counts alike, so the segments tile the range.
//   Then the ranges are written and every wave sorts its TPW tiles in place (rank_sort, or the wave-owned network of
//   k_sort_tiles_small, from LDS instead of global memory; up to 64 keys: ranks by counting) and writes the ids.
//   A tile of more than AREA instances (rare where this kernel is chosen) is staged as plain keys in global scratch
//   by a second pass over the bucket and sorted by the whole workgroup with the long-tile routines above.
// Chosen by the host from sizes alone (binning_plan): both this kernel and the chain are correct for every input.
#define BSR_BKT_NT 512
#define BSR_BKT_NW (BSR_BKT_NT / 64)
typedef uint32_t bsr_u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <int AREA, int TPW>   // keys per tile area (512 / 1024 / 2048); tiles per wave
__global__ void __launch_bounds__(BSR_BKT_NT) k_bucket_sort(int T, int k_log2, const int* __restrict__ n_ptr, int capacity,
                                                            const uint32_t* __restrict__ digit_total1,
                                                            const BinElem* __restrict__ elems, uint2* __restrict__ tile_range,
                                                            uint64_t* big_keys, uint32_t* __restrict__ point_list,
                                                            int sort_mode)
{
	const int force_int = sort_mode & 1;
	constexpr int NT = BSR_BKT_NT, NW = BSR_BKT_NW, NA = NW * TPW;
	// TPW = 2: a wave sorts its two tiles side by side, one per 32-lane half, 16 keys per lane and round (slots swz_m<4>);
	// TPW = 1: one tile per wave, 8 keys per lane and round (slots swz_m<3>)
	constexpr int SM = TPW == 2 ? 4 : 3;
	static_assert(TPW == 1 || (TPW == 2 && AREA == 512), "paired sort: two 512-key areas per wave");
	static_assert(NA * AREA >= BSR_SORT_CHUNK, "the long-tile routines sort 4096-key chunks in this LDS");
	__shared__ uint64_t s_keys[NA * AREA];          // one area per owned tile; the long-tile routines use the first 4096 slots
	__shared__ uint32_t s_cnt[NA];                  // elements per owned tile (the fill counters of the pass)
	__shared__ uint32_t s_part[4];                  // [0]: elements of the bucket in earlier parts
	__shared__ uint32_t s_cur[NA];                  // second pass: fill counters of this part's long tiles
	// rank_sort's counters: 512 buckets per wave, 1024 where an area holds 2048 keys (the long-tile routine: all of it)
	constexpr int RNBLOG = AREA > 1024 ? 10 : 9, RDW = (1 << RNBLOG) / 2;
	__shared__ uint32_t s_rank[NW][RDW];
	static_assert(NW * RDW >= 1024 + 32, "sort_long_tile_lds takes 2048 buckets + its reduction words");
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int k = 1 << k_log2;                      // 1, 2 or 4
	// parts of one bucket are neighbours on one XCD: workgroups b, b + 8, b + 16, ... share an XCD
	const int xcd = (int)blockIdx.x & 7, r = (int)blockIdx.x >> 3;
	const int j = r & (k - 1), d = ((r >> k_log2) << 3) | xcd;
	const int nt = d < T ? ((T - 1 - d) >> BSR_RADIX_BITS) + 1 : 0;      // tiles of this bucket: (hi << 8) | d < T
	const int m = nt > j ? (nt - j + k - 1) >> k_log2 : 0;               // ... of this part: hi = j + k L, L < m <= NA
	const int n_all = *n_ptr;
	if (n_all > capacity) return;   // scratch too small: the stage is re-run
	if (n_all <= 0) {               // nothing kept: every tile is empty
		if (tid < m) tile_range[(uint32_t)((j + (tid << k_log2)) << BSR_RADIX_BITS) | (uint32_t)d] = make_uint2(0u, 0u);
		return;
	}
	// the bucket's range: its base among the 256 digits (written by k_emit_scatter behind the totals) and its total
	if (tid < NA) s_cnt[tid] = 0u;
	if (tid < 4) s_part[tid] = 0u;
	__syncthreads();
	const uint32_t beg = digit_total1[BSR_RADIX_BINS + d], size = digit_total1[d];   // (uniform: scalar loads)
	const uint2* const src = reinterpret_cast<const uint2*>(elems) + beg;
	// An element of the pass is counted if it belongs to an EARLIER part (all the layout needs of the other parts: where
	// this part's tiles begin -- one vote per element; until the 2048-key areas every part was counted, k votes); if it
	// belongs to a tile of this part it takes its slot in the tile's area from the tile's LDS counter.  Eight elements at
	// a time: first all eight returning atomics, then the eight stores -- element by element every atomic's round trip
	// through the LDS was waited for before the next element was looked at (16 round trips per trip of a wave).
	uint32_t before = 0u;   // (wave-uniform: a scalar register)
	auto take8 = [&](const bool full, const uint32_t i, const bsr_u32x4_a8 (&v)[8], const int u0) {
		uint32_t pos[8];
#pragma unroll
		for (int e = 0; e < 8; e++) {
			const bsr_u32x4_a8 q = v[u0 + (e >> 1)];
			const uint32_t w0 = (e & 1) ? q.z : q.x;
			const uint32_t ie = i + 2u * (uint32_t)((e >> 1) * NT) + (uint32_t)(e & 1);
			const bool valid = full || ie < size;
			const uint32_t hi = w0 >> 24;
			const uint32_t part = hi & (uint32_t)(k - 1);
			if (j > 0) {   // (j: workgroup-uniform.  Two votes AND-ed on the scalar side: a vote on `valid && ...` is a mask
				           // materialised in a VGPR and compared again)
				const uint64_t m = wave_ballot(part < (uint32_t)j);
				before += (uint32_t)__popcll(full ? m : (m & wave_ballot(valid)));
			}
			pos[e] = 0xffffffffu;
			if (valid && (int)part == j) pos[e] = atomicAdd(&s_cnt[hi >> k_log2], 1u);   // LDS
		}
#pragma unroll
		for (int e = 0; e < 8; e++) {
			const bsr_u32x4_a8 q = v[u0 + (e >> 1)];
			const uint32_t w0 = (e & 1) ? q.z : q.x, w1 = (e & 1) ? q.w : q.y;
			if (pos[e] < (uint32_t)AREA)
				s_keys[(w0 >> (24 + k_log2)) * AREA + swz_m<SM>((int)pos[e])] = ((uint64_t)w1 << 32) | (uint64_t)(w0 & 0x00ffffffu);
		}
	};
	// ---- the pass over the bucket: two elements per 16-byte load, eight loads in flight
	auto request = [&](uint32_t i0, bsr_u32x4_a8 (&v)[8]) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			const uint32_t i = i0 + 2u * (uint32_t)(u * NT + tid);
			if (i + 1 < size) v[u] = *reinterpret_cast<const bsr_u32x4_a8*>(src + i);
			else if (i < size) { const uint2 e = src[i]; v[u] = bsr_u32x4_a8{e.x, e.y, 0u, 0u}; }
			else v[u] = bsr_u32x4_a8{0u, 0u, 0u, 0u};
		}
	};
	auto consume = [&](uint32_t i0, const bsr_u32x4_a8 (&v)[8]) {
		const bool full = i0 + (uint32_t)(NT * 16) <= size;   // (uniform) every element of the trip exists
		const uint32_t i = i0 + 2u * (uint32_t)tid;
		take8(full, i, v, 0);
		if (i0 + 2u * (uint32_t)(4 * NT) < size) take8(full, i + 2u * (uint32_t)(4 * NT), v, 4);   // (uniform bound)
	};
	// (k_bucket_sort<2048, 1>, one workgroup per CU: requesting the next trip's loads before this one's elements are taken
	// -- two register sets -- was measured: 130 us against 117; sixteen waves of which eight sort: 128)
	for (uint32_t i0 = 0; i0 < size; i0 += NT * 16) {
		bsr_u32x4_a8 v[8];
		request(i0, v);
		consume(i0, v);
	}
	if (lane == 0 && before != 0u) atomicAdd(&s_part[0], before);   // LDS
	__syncthreads();
	// ---- layout: part-major inside the bucket's range, tiles of a part in order of L
	const uint32_t part_beg = beg + s_part[0];
	// every wave keeps the owned tiles' counts and first positions in its lanes (lane L <-> tile L): one LDS read and a
	// DPP scan instead of a serial sum of up to 16 counters per look-up
	static_assert(NA <= 64, "one lane per owned tile");
	const uint32_t cnt_lane = lane < NA ? s_cnt[lane] : 0u;
	const uint32_t first_lane = part_beg + wave_inclusive_sum_dpp(cnt_lane) - cnt_lane;
	auto tile_first = [&](int L) {   // first position of owned tile L (L wave-uniform)
		return (uint32_t)__builtin_amdgcn_readlane((int)first_lane, L);
	};
	auto tile_first_any = [&](int L) {   // the same for a lane's own L (the long-tile pass)
		uint32_t f = part_beg;
		for (int q = 0; q < L; q++) f += s_cnt[q];
		return f;
	};
	if (tid < m) tile_range[(uint32_t)((j + (tid << k_log2)) << BSR_RADIX_BITS) | (uint32_t)d] = make_uint2(first_lane, first_lane + cnt_lane);
	// ---- every wave sorts its tiles
	const bool any_long = wave_ballot(lane < m && cnt_lane > (uint32_t)AREA) != 0ull;   // (workgroup-uniform: every wave sees all counts)
	// ranks by counting, tiles of up to 64 keys ((depth bits, id) pairs are unique within a tile): no network, no further
	// LDS traffic
	auto sort_by_ranks = [&](const uint64_t* keys, int n, uint32_t start) {
		uint64_t key = ~0ull;
		if (lane < n) key = keys[swz_m<SM>(lane)];
		const uint32_t kh = (uint32_t)(key >> 32), kl = (uint32_t)key;
		uint32_t rank = 0;
		for (int q = 0; q < n; q++) {
			const uint64_t kq = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)kh, q) << 32) |
			                    (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)kl, q);
			rank += kq < key ? 1u : 0u;
		}
		if (lane < n) point_list[start + rank] = kl;
	};
	if constexpr (TPW == 2) {
		// tiles L = wave (lanes 0..31) and wave + NW (lanes 32..63), the same schedule for both: n2 = the larger one's
		const int LA = wave, LB = wave + NW;
		int nA = LA < m ? (int)s_cnt[LA] : 0, nB = LB < m ? (int)s_cnt[LB] : 0;
		if (nA > AREA) nA = 0;   // (long: sorted further down)
		if (nB > AREA) nB = 0;
		const uint32_t startA = tile_first(LA < m ? LA : 0), startB = tile_first(LB < m ? LB : 0);
		// bucket-and-rank sort, one tile after the other with all 64 lanes (sort_mode 0); a tile it declines stays as it
		// is and goes through the network below
		if (sort_mode == 0) {
			auto by_ranks = [&](int L, int& n, uint32_t start) {
				if (n <= 64) return;
				uint64_t* const keys = s_keys + L * AREA;
				uint64_t e[AREA / 64];
#pragma unroll
				for (int q = 0; q < AREA / 64; q++) e[q] = lane + 64 * q < n ? keys[swz_m<SM>(lane + 64 * q)] : 0ull;
				round_sync<false>();
				if (rank_sort<64, AREA / 64, RNBLOG, false>(e, n, lane, keys, s_rank[wave], nullptr, start, point_list)) n = 0;
				round_sync<false>();
			};
			by_ranks(LA, nA, startA);
			by_ranks(LB, nB, startB);
		}
		if (nA <= 64 && nB <= 64) {
			if (nA > 0) sort_by_ranks(s_keys + LA * AREA, nA, startA);
			if (nB > 0) sort_by_ranks(s_keys + LB * AREA, nB, startB);
		} else {
			int n2 = 128;
			while (n2 < nA || n2 < nB) n2 <<= 1;
			const int half = lane >> 5, t = lane & 31;
			const int n = half ? nB : nA;
			const uint32_t start = half ? startB : startA;
			uint64_t* const keys = s_keys + (half ? LB : LA) * AREA;
			// pads, then runs of 16 sorted in registers, in place: a lane reads and writes the same sixteen slots
			round_sync<false>();
			for (int i = n + t; i < n2; i += 32) keys[swz_m<4>(i)] = BSR_PAD_KEY;
			round_sync<false>();
			bool plain = true;
			for (int i = t * 16; i < n2; i += 32 * 16) {
				uint64_t e[16];
				const int p0 = swz_m<4>(i);
#pragma unroll
				for (int q = 0; q < 16; q++) {
					e[q] = keys[p0 ^ swz_m<4>(q)];
					plain = plain && (i + q >= n || key_is_plain_double(e[q]));
				}
				reg_sort<4, false>(e);
#pragma unroll
				for (int q = 0; q < 16; q++) keys[p0 ^ swz_m<4>(q)] = e[q];
			}
			if (wave_ballot(!(plain && !force_int)) == 0ull)
				merge_loaded_runs<32, 4, false, true>(keys, n2, start, n, t, point_list);
			else
				merge_loaded_runs<32, 4, false, false>(keys, n2, start, n, t, point_list);
		}
	} else
	for (int L = wave; L < m; L += NW) {
		const int n = (int)s_cnt[L];
		const uint32_t start = tile_first(L);
		uint64_t* const keys = s_keys + L * AREA;
		if (n > 0 && n <= 64) {
			sort_by_ranks(keys, n, start);
		} else if (n > 64 && n <= AREA) {
			if (sort_mode == 0) {   // bucket-and-rank sort; a tile it declines goes through the network
				uint64_t e[AREA / 64];
#pragma unroll
				for (int q = 0; q < AREA / 64; q++) e[q] = lane + 64 * q < n ? keys[swz_m<SM>(lane + 64 * q)] : 0ull;
				round_sync<false>();
				const bool done = rank_sort<64, AREA / 64, RNBLOG, false>(e, n, lane, keys, s_rank[wave], nullptr, start, point_list);
				round_sync<false>();
				if (done) continue;
			}
			int n2 = 128;
			while (n2 < n) n2 <<= 1;
			// pads, then runs of 8 sorted in registers, in place: a lane reads and writes the same eight slots
			round_sync<false>();
			for (int i = n + lane; i < n2; i += 64) keys[swz_m<3>(i)] = BSR_PAD_KEY;
			round_sync<false>();
			bool plain = true;
			for (int i = lane * 8; i < n2; i += 64 * 8) {
				uint64_t e[8];
				const int p0 = swz_m<3>(i);
#pragma unroll
				for (int q = 0; q < 8; q++) {
					e[q] = keys[p0 ^ swz_m<3>(q)];
					plain = plain && (i + q >= n || key_is_plain_double(e[q]));
				}
				reg_sort<3, false>(e);
#pragma unroll
				for (int q = 0; q < 8; q++) keys[p0 ^ swz_m<3>(q)] = e[q];
			}
			if (wave_ballot(!(plain && !force_int)) == 0ull)
				merge_loaded_runs<64, 3, false, true>(keys, n2, start, n, lane, point_list);
			else
				merge_loaded_runs<64, 3, false, false>(keys, n2, start, n, lane, point_list);
		}
	}
	if (!any_long) return;
	// ---- long tiles of this part: a second pass over the bucket stages their keys in global scratch, at the segment's
	// own positions; then the whole workgroup sorts them one by one
	__syncthreads();
	if (tid < NA) s_cur[tid] = 0u;
	__syncthreads();
	for (uint32_t i0 = 0; i0 < size; i0 += NT * 4) {
		uint2 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const uint32_t i = i0 + (uint32_t)(u * NT + tid);
			v[u] = i < size ? src[i] : make_uint2(0u, 0u);
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const uint32_t i = i0 + (uint32_t)(u * NT + tid);
			if (i < size) {
				const uint32_t hi = v[u].x >> 24;
				const uint32_t L = hi >> k_log2;
				if ((int)(hi & (uint32_t)(k - 1)) == j && s_cnt[L] > (uint32_t)AREA) {
					const uint32_t pos = atomicAdd(&s_cur[L], 1u);   // LDS
					big_keys[(size_t)tile_first_any((int)L) + pos] = ((uint64_t)v[u].y << 32) | (uint64_t)(v[u].x & 0x00ffffffu);
				}
			}
		}
	}
	__threadfence();
	__syncthreads();
	const RawKeys raw{big_keys};
	for (int L = 0; L < m; L++) {
		const int n = (int)s_cnt[L];
		if (n <= AREA) continue;   // (uniform)
		const uint32_t start = tile_first(L);
		if (n <= BSR_SORT_CHUNK)
			sort_long_tile_lds<NT>(s_keys, &s_rank[0][0], start, n, tid, raw, point_list, sort_mode);
		else
			sort_long_tile_hybrid<NT>(s_keys, big_keys + start, start, n, tid, raw, point_list);
	}
}

// once per forward call, right after k_preprocess (independent of the instance count: runs before the read-back)
void launch_scans(int n_wg, uint32_t* wg_kept, uint32_t* wg_area, int* flags, uint32_t* hist1, int* host_counts,
                  hipStream_t s)
{
	hipLaunchKernelGGL(k_scans, dim3(BSR_RADIX_BINS + 1), dim3(1024), 0, s, n_wg, wg_kept, wg_area, flags, hist1,
	                   host_counts);
}

// Which second pass a forward call runs -- decided on the host from sizes alone, every plan is correct for every input:
//   0  tile ids beyond 16 bits (stacked views, > 4096 x 4096): the remaining LSD radix passes + k_tile_ranges,
//   1  tile-owned chain: k_tile_count -> k_tile_starts -> k_tile_scatter, then the per-tile sort launches,
//   2  k_bucket_sort<1024, 1> (one launch for the second pass AND the sort): up to 8192 tiles, Gaussian ids below 2^24,
//      and lists that are short on average -- a workgroup streams its whole bucket, up to four workgroups per bucket:
//      that pays while an average tile holds well under the 1024 keys of a tile area (C3: 366 kept instances per tile;
//      the dense leg: 1600, C5: 1850),
//   3  k_bucket_sort<512, 2>: the same with 512-key areas, two tiles per wave (half the workgroups per bucket), where
//      the average tile holds at most BSR_BKT_SMALL_PER_TILE,
//   4  k_bucket_sort<2048, 1>: 2048-key areas (128 KB of keys: one workgroup per CU, four workgroups per bucket of a
//      1080p frame) for dense frames, up to BSR_BKT_BIG_PER_TILE per tile on average: against the chain it saves the
//      count, the global scan and the scatter (the elements written and read once more) for three more reads of the
//      bucket from the L2.
// kept_hint = the number of kept instances the caller expects (this frame's count when the host has read it, the
// previous frame's while it guesses), 0 = unknown: 70 % of the scratch capacity then (the exact tile cull keeps ~2/3
// of the reference's instances on the synthetic scenes).
// The debug builds libbsr_chain_only.so / libbsr_bucket_always.so (csrc/Makefile) pin plan 1 / 2-3 for the tests.
#ifndef BSR_BUCKET_MAX_PER_TILE
#define BSR_BUCKET_MAX_PER_TILE 850   // (A/B on one box, both forms forced: 512 x 512 with ~750 per tile +2.8 % of the step with
                                      // the bucket form; the dense leg, 1200 per tile -- most tiles past their area -- -21 %)
#endif
#ifndef BSR_BKT_SMALL_PER_TILE
#define BSR_BKT_SMALL_PER_TILE 400
#endif
#ifndef BSR_BKT_BIG_PER_TILE
#define BSR_BKT_BIG_PER_TILE 1400   // up to here: 2048-key areas (plan 4), one workgroup of 144 KB per CU.  (A/B, all forms forced:
                                    // dense leg, 1213 per tile: second pass + sorts 216 us against the chain's 240; 925 per tile:
                                    // even; C5, 1830 per tile with tiles past 2048: even, and 20 % slower at 2000 per tile)
#endif
int binning_plan(int P, int T, int capacity, long long kept_hint)
{
	int bits = 0;
	while ((1 << bits) < T) bits++;
	const bool tile_owned = bits <= 2 * BSR_RADIX_BITS && 2 * (size_t)T <= (size_t)BSR_RADIX_BINS * BSR_HIST_BLOCKS_MAX;
	if (!tile_owned) return 0;
	const long long kept = kept_hint > 0 ? kept_hint : (long long)capacity * 7 / 10;
	if (P <= (1 << 24) && T <= 8192 && kept <= (long long)BSR_BUCKET_MAX_PER_TILE * T)
		return kept <= (long long)BSR_BKT_SMALL_PER_TILE * T ? 3 : 2;
	if (P <= (1 << 24) && T <= 8192 && BSR_BUCKET_MAX_PER_TILE > 0 && kept <= (long long)BSR_BKT_BIG_PER_TILE * T) return 4;
	return 1;
}

// Bins the kept instances (their number is read from *n_ptr on the device): emit -> second pass on the tile id -> tile
// ranges (plan 2: the second pass is part of launch_sort_tiles).  elems_a / elems_b ping-pong; *elems_sorted is the
// buffer the sort stage reads.  Grids are sized for `capacity` instances; workgroups beyond the real count exit.
void launch_binning(int plan, int P, int T, int gx, const int* n_ptr, int capacity, const GeomState& geom, BinElem* elems_a,
                    BinElem* elems_b, uint32_t* hist, int hist_blocks_max, uint2* tile_range, uint32_t* big_tiles,
                    int* flags, BinElem** elems_sorted, BinElem** elems_free, int* compact_out, hipStream_t s)
{
	// pass 1 (tile id bits 0..7) fused with the emit; geom.hist1 was row-scanned by launch_scans
	int bits = 0;
	while ((1 << bits) < T) bits++;
	const bool tile_owned = plan != 0;
	// 8-byte elements where the tile-owned pass runs and Gaussian ids fit 24 bits (common.h: load_elem_m)
	const int compact = (tile_owned && P <= (1 << 24)) ? 1 : 0;
	*compact_out = compact;
	uint32_t* tile_count = hist;          // [T]   (the histogram area of the generic passes, unused on this path)
	uint32_t* tile_cursor = hist + T;     // [T]
	hipLaunchKernelGGL(k_emit_scatter, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, n_ptr, capacity, geom.rect,
	                   geom.kept_mask, geom.depth, geom.hist1, elems_a, tile_count, plan == 1 ? 2 * T : 0, compact);
	if (plan >= 2) {   // the bucket-owned second pass is fused with the sort (launch_sort_tiles)
		*elems_sorted = elems_a;
		*elems_free = elems_b;
		return;
	}
	if (tile_owned) {
		// tile ids of up to 16 bits (every single-view call up to 4096 x 4096): count -> starts -> scatter
		const int n_wg = (P + 255) / 256, n_col = 8 * ((n_wg + 7) >> 3);
		const uint32_t* digit_total1 = geom.hist1 + (size_t)BSR_RADIX_BINS * n_col;
		// slices of ~3000 elements at full capacity: k_tile_scatter holds up to 4096 in registers (one read of the
		// slice), and the global atomics stay at a few per hundred elements
		int n_slices = capacity / (BSR_RADIX_BINS * 3072) + 1;
		n_slices = n_slices > 256 ? 256 : n_slices;
		hipLaunchKernelGGL(k_tile_count, dim3(BSR_RADIX_BINS * n_slices), dim3(256), 0, s, T, n_slices, n_ptr, capacity,
		                   digit_total1, elems_a, tile_count, compact);
		if (T <= 8192)
			hipLaunchKernelGGL(k_tile_starts<8>, dim3(1), dim3(1024), 0, s, T, n_ptr, capacity, tile_count, tile_range,
			                   big_tiles, flags);
		else
			hipLaunchKernelGGL(k_tile_starts<64>, dim3(1), dim3(1024), 0, s, T, n_ptr, capacity, tile_count, tile_range,
			                   big_tiles, flags);
		hipLaunchKernelGGL(k_tile_scatter, dim3(BSR_RADIX_BINS * n_slices), dim3(256), 0, s, T, n_slices, n_ptr, capacity,
		                   digit_total1, elems_a, elems_b, tile_range, tile_cursor, compact);
		*elems_sorted = elems_b;
		*elems_free = elems_a;
		return;
	}
	int max_blocks = (capacity + 1023) / 1024;   // chunk >= 1024
	if (max_blocks > hist_blocks_max) max_blocks = hist_blocks_max;
	if (max_blocks < 1) max_blocks = 1;
	uint32_t* digit_total = hist + (size_t)BSR_RADIX_BINS * hist_blocks_max;
	BinElem* ei = elems_a; BinElem* eo = elems_b;
	for (int shift = BSR_RADIX_BITS; shift < bits; shift += BSR_RADIX_BITS) {
		hipLaunchKernelGGL(k_radix_hist, dim3(max_blocks), dim3(256), 0, s, n_ptr, capacity, hist_blocks_max, shift, ei,
		                   hist);
		hipLaunchKernelGGL(k_radix_rowscan, dim3(BSR_RADIX_BINS), dim3(256), 0, s, n_ptr, capacity, hist_blocks_max, hist,
		                   digit_total);
		hipLaunchKernelGGL(k_radix_scatter, dim3(max_blocks), dim3(256), 0, s, n_ptr, capacity, hist_blocks_max, shift, ei,
		                   eo, hist, digit_total);
		BinElem* tt = ei; ei = eo; eo = tt;
	}
	// one wave per three tiles
	hipLaunchKernelGGL(k_tile_ranges, dim3((T / 3 + 1 + 3) / 4), dim3(256), 0, s, T, n_ptr, capacity, ei, tile_range,
	                   big_tiles, flags);
	*elems_sorted = ei;
	*elems_free = eo;
}

// Size classes: (0, 1024] -> one WAVE per tile (8 KB of LDS each); the wide classes share ONE launch
// (k_sort_tiles_wide): (1024, 4096] sorted in 32 KB of LDS, longer segments hybrid in 4096-key chunks, one entry of a
// work list per workgroup.  n instances can fill at most n / 1024 (n / 4096) such tiles, which bounds the grid: a frame
// without long lists pays one near-empty launch, not 3 x T idle workgroups.
void launch_sort_tiles(int plan, int T, int n_bound, const int* n_ptr, int capacity, uint2* tile_range,
                       const uint32_t* big_tiles, const int* flags, const uint32_t* digit_total1, const BinElem* elems,
                       BinElem* elems_free, uint32_t* point_list, int compact, int force_int, int small_grids_flag,
                       hipStream_t s)
{
	// force_int = the call's sort mode: bit 0 BSR_FLAG_TEST_SORT_INT, bit 1 BSR_FLAG_TEST_SORT_NETWORK (0: bucket-and-rank
	// sort first, the network for the segments it declines); small_grids_flag: BSR_FLAG_TEST_SMALL_GRIDS (test-only, no result
	// changes): every segment through the integer compare-exchange flavour, which real inputs reach only with NaN /
	// non-positive depth bits; the wide classes on grids of 2 / 1 workgroups
	if (plan >= 2) {
		// second pass + sort in one launch: `elems` is still in pass-1 order; the free buffer holds the keys of long tiles.
		// 64 KB of LDS either way: 16 tile areas of 512 keys (two per wave) or 8 of 1024; a bucket has ceil(T / 256)
		// tiles, split over k = 1, 2 or 4 workgroups.
		const int nt_max = (T + BSR_RADIX_BINS - 1) / BSR_RADIX_BINS;
		const bool small_areas = plan == 3, big_areas = plan == 4;
		const int per_wg = small_areas ? 2 * BSR_BKT_NW : BSR_BKT_NW;
		int k_log2 = 0;
		while ((per_wg << k_log2) < nt_max) k_log2++;
		uint64_t* const big_keys = reinterpret_cast<uint64_t*>(elems_free);
		if (small_areas)
			hipLaunchKernelGGL((k_bucket_sort<512, 2>), dim3(BSR_RADIX_BINS << k_log2), dim3(BSR_BKT_NT), 0, s, T, k_log2, n_ptr,
			                   capacity, digit_total1, elems, tile_range, big_keys, point_list, force_int);
		else if (big_areas)
			hipLaunchKernelGGL((k_bucket_sort<2048, 1>), dim3(BSR_RADIX_BINS << k_log2), dim3(BSR_BKT_NT), 0, s, T, k_log2, n_ptr,
			                   capacity, digit_total1, elems, tile_range, big_keys, point_list, force_int);
		else
			hipLaunchKernelGGL((k_bucket_sort<1024, 1>), dim3(BSR_RADIX_BINS << k_log2), dim3(BSR_BKT_NT), 0, s, T, k_log2, n_ptr,
			                   capacity, digit_total1, elems, tile_range, big_keys, point_list, force_int);
		return;
	}
	// sparse frames (the views of a camera sweep: every tile a few dozen entries) get the tiny class its own kernel;
	// where tiles average 128 entries or more the few short ones stay with the small class (one launch fewer)
	const bool tiny = (long long)n_bound < 128ll * T;
	if (tiny)
		hipLaunchKernelGGL(k_sort_tiles_tiny, dim3((T + 3) / 4), dim3(256), 0, s, T, n_ptr, capacity, tile_range, elems,
		                   point_list, compact);
	hipLaunchKernelGGL(k_sort_tiles_small, dim3((T + 3) / 4), dim3(256), 0, s, T, n_ptr, capacity, tile_range, elems, point_list,
	                   force_int, tiny ? 64 : 0, compact);
	// n instances can fill at most n / 1025 tiles of the first wide class and n / 4097 of the two longer ones: the grid
	// covers both work lists (n_bound >= the real count), capped -- the workgroups stride over their lists
	// (BSR_FLAG_TEST_SMALL_GRIDS: caps of 2 / 1, so that ordinary test frames drive several tiles through one
	// workgroup's striding loop -- with the product caps that takes > 2560 / 512 long tiles in one frame)
	const bool small_grids = small_grids_flag != 0;
	const int g1 = min(min(T, n_bound / (BSR_SORT_SMALL + 1)), small_grids ? 2 : 2560),
	          gw = min(min(T, n_bound / (BSR_SORT_CHUNK + 1)), small_grids ? 1 : 512);
	// (1024, 2048] in a launch of its own where the frame can hold a fair number of such tiles (a dense frame); a frame
	// of short lists pays no second near-empty launch.  The first wide class then starts at 2049 keys: at most
	// n / 2049 such tiles, and its near-empty launch is kept small (2560 workgroups of 512 that find nothing: 10 us).
	const bool mid = g1 >= 64 || small_grids;
	int g1w = g1;
	if (mid) {
		const int gm = small_grids ? 2 : min(g1, BSR_SORT_MID_WGS);
		hipLaunchKernelGGL(k_sort_tiles_mid, dim3(gm), dim3(256), 0, s, gm, n_ptr, capacity, tile_range, big_tiles, flags, elems,
		                   point_list, force_int, compact);
		g1w = min(min(T, n_bound / (BSR_SORT_MID + 1)), small_grids ? 2 : 640);
	}
	if (g1w + gw > 0)
		hipLaunchKernelGGL(k_sort_tiles_wide, dim3(g1w + gw), dim3(BSR_SORT_NT), 0, s, T, g1w, n_ptr, capacity, tile_range,
		                   big_tiles, flags, elems, reinterpret_cast<uint64_t*>(elems_free), point_list, force_int, compact,
		                   mid ? BSR_SORT_MID : BSR_SORT_SMALL);
}

}  // namespace bsr
