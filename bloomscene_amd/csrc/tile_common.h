// Pieces shared by the forward and backward tile renderers: the conservative quadrant test and
// the ballot/prefix-scan compaction of a staged batch into one ordered list per wave.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace bsr {

// Measurement hook of the occupancy sweep (BASELINE config C5: "LDS-tile occupancy + rocprof HBM-GB/s sweep",
// tools/sweep_occupancy.sh): extra dynamic LDS bytes per workgroup of a tile renderer, unused by the kernel, which
// only lowers the number of workgroups a CU can hold.  Read once from the environment; 0 (unset) in normal use.
static inline unsigned occupancy_sweep_lds_pad(const char* env_name)
{
	const char* v = getenv(env_name);
	return v ? (unsigned)strtoul(v, nullptr, 10) : 0u;
}

// Can the splat reach alpha >= 1/255 at ANY point of an axis-aligned box of pixel centres
// [bx, bx+EXT] x [by, by+EXT]?  power(d) = -q(d), q(d) = 0.5*(a dx^2 + c dy^2) + b dx dy with
// d = centre - pixel.  For a positive-definite conic q is convex, so its minimum over the box is 0
// if the centre lies inside and otherwise sits on one of the four edges, where it is a clamped 1-D
// parabola minimum.  The splat is kept iff  -qmin >= power_cut - slack  (power_cut already carries
// a margin; the extra slack covers the rounding of this test).  Anything not provably a miss --
// non-PD conics, NaNs -- is kept, so the per-pixel decisions downstream stay exact.
template <int EXT, int EXTY = EXT>   // box of pixel centres [bx, bx+EXT] x [by, by+EXTY]: EXT = 7 (quadrant) or 15 (tile)
__device__ __forceinline__ bool box_may_hit(float X, float Y, float a, float b, float c, float cut, float rb_c,
                                            float rb_a, bool pd, float bx, float by)
{
	const float dx_lo = X - (bx + (float)EXT), dx_hi = X - bx;
	const float dy_lo = Y - (by + (float)EXTY), dy_hi = Y - by;
	const bool in_x = (dx_lo <= 0.0f) && (dx_hi >= 0.0f);
	const bool in_y = (dy_lo <= 0.0f) && (dy_hi >= 0.0f);
	float qmin = 0.0f;
	if (!(in_x && in_y)) {
		// edges dx = const
		float dy0 = fminf(fmaxf(rb_c * dx_lo, dy_lo), dy_hi);
		float q0 = 0.5f * (a * dx_lo * dx_lo + c * dy0 * dy0) + b * dx_lo * dy0;
		float dy1 = fminf(fmaxf(rb_c * dx_hi, dy_lo), dy_hi);
		float q1 = 0.5f * (a * dx_hi * dx_hi + c * dy1 * dy1) + b * dx_hi * dy1;
		// edges dy = const
		float dx2 = fminf(fmaxf(rb_a * dy_lo, dx_lo), dx_hi);
		float q2 = 0.5f * (a * dx2 * dx2 + c * dy_lo * dy_lo) + b * dx2 * dy_lo;
		float dx3 = fminf(fmaxf(rb_a * dy_hi, dx_lo), dx_hi);
		float q3 = 0.5f * (a * dx3 * dx3 + c * dy_hi * dy_hi) + b * dx3 * dy_hi;
		qmin = fminf(fminf(q0, q1), fminf(q2, q3));
	}
	const float slack = 1.0e-3f + 1.0e-4f * fabsf(cut);
	const bool miss = pd && (-qmin < cut - slack);
	return !miss;
}

// Staged batch of up to BATCH (<= 256) list entries + per-wave (= per 8x8 quadrant) compacted
// index lists.  The forward stages 256 entries at a time; the backward 128, which with its
// per-wave partial-sum slots keeps LDS at 25 KB per workgroup (6 workgroups per CU).
template <int BATCH>
struct TileStageT {
	float4 q0[BATCH];                // x, y, conic a, conic b
	float4 q1[BATCH];                // conic c, power cut, opacity, depth
	float4 q2[BATCH];                // r, g, b, -
	unsigned int list[4][BATCH];     // per quadrant: BYTE offsets (entry << 4) into q0 / q1 / q2, in list order
	                                 // (32-bit: the walks fetch four entries with one 16-byte read, no unpacking)
	unsigned int cnt[4][4];          // [staging wave][quadrant]
};
using TileStage = TileStageT<BSR_BLOCK>;

// Thread `tid` holds the record of batch entry `tid` (valid iff tid < cnt).  Writes the record
// to LDS and appends tid << 4 (its byte offset in the three record arrays, so the walks use the list
// value as the LDS address directly) to the list of every quadrant it may touch, preserving list order
// (64-bit ballots + prefix popcounts inside a wave, a 4x4 count table across waves).
// On return (after the trailing barrier) st.list[q][0 .. total[q]) is ready; returns total[wave].
template <int BATCH>
__device__ __forceinline__ int stage_and_compact(TileStageT<BATCH>& st, int tid, bool valid, const float4 r0,
                                                 const float4 r1, const float4 r2, float tile_x0, float tile_y0)
{
	const int wave = tid >> 6, lane = tid & 63;
	bool h[4] = {false, false, false, false};
	if (valid) {
		st.q0[tid] = r0;
		st.q1[tid] = r1;
		st.q2[tid] = r2;
		const float a = r0.z, b = r0.w, c = r1.x;
		const bool pd = (a > 0.0f) && (c > 0.0f) && (a * c - b * b > 0.0f);
		const float rb_c = -b / c, rb_a = -b / a;
#pragma unroll
		for (int q = 0; q < 4; q++)
			h[q] = box_may_hit<7>(r0.x, r0.y, a, b, c, r1.y, rb_c, rb_a, pd, tile_x0 + (float)((q & 1) << 3),
			                    tile_y0 + (float)((q >> 1) << 3));
	}
	unsigned long long m[4];
#pragma unroll
	for (int q = 0; q < 4; q++) m[q] = wave_ballot(h[q]);
	if (lane == 0) {
#pragma unroll
		for (int q = 0; q < 4; q++) st.cnt[wave][q] = (unsigned int)__popcll(m[q]);
	}
	__syncthreads();
	const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
	for (int q = 0; q < 4; q++) {
		if (h[q]) {
			const unsigned int off = (wave > 0 ? st.cnt[0][q] : 0u) + (wave > 1 ? st.cnt[1][q] : 0u) +
			                         (wave > 2 ? st.cnt[2][q] : 0u);
			st.list[q][off + (unsigned int)__popcll(m[q] & lt)] = (unsigned int)(tid << 4);
		}
	}
	const int total = (int)(st.cnt[0][wave] + st.cnt[1][wave] + st.cnt[2][wave] + st.cnt[3][wave]);
	__syncthreads();
	return total;
}

// Record fields of the list entry at byte offset `joff` (wave-uniform, but deliberately left in a VGPR:
// a readfirstlane + scalar shift + move back costs four issue slots per visit and buys nothing, the
// LDS broadcasts a uniform address anyway).
template <int BATCH>
__device__ __forceinline__ const char* stage_rec(const TileStageT<BATCH>& st, unsigned int joff)
{
	return reinterpret_cast<const char*>(&st.q0[0]) + joff;
}
template <int BATCH> __device__ __forceinline__ float4 rec_q0(const char* r) { return *reinterpret_cast<const float4*>(r); }
template <int BATCH> __device__ __forceinline__ float4 rec_q1(const char* r) { return *reinterpret_cast<const float4*>(r + BATCH * 16); }
template <int BATCH> __device__ __forceinline__ float2 rec_q1lo(const char* r) { return *reinterpret_cast<const float2*>(r + BATCH * 16); }
template <int BATCH> __device__ __forceinline__ float2 rec_q1hi(const char* r) { return *reinterpret_cast<const float2*>(r + BATCH * 16 + 8); }
template <int BATCH> __device__ __forceinline__ float4 rec_q2(const char* r) { return *reinterpret_cast<const float4*>(r + BATCH * 32); }

}  // namespace bsr
