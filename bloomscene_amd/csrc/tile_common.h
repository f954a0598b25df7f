// Pieces shared by the forward and backward tile renderers: the conservative quadrant test and
// the ballot/prefix-scan compaction of a staged batch into one ordered list per wave.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace bsr {

// Measurement hook of the occupancy sweep (BASELINE config C5: "LDS-tile occupancy + rocprof HBM-GB/s sweep",
// tools/sweep_occupancy.sh): extra dynamic LDS bytes per workgroup of a tile renderer, unused by the kernel, which
// only lowers the number of workgroups a CU can hold.  Compiled in only by the sweep build (make sweep ->
// libbsr_rast_sweep.so, -DBSR_OCCUPANCY_SWEEP); the product library reads no environment variable on its launch paths.
#ifdef BSR_OCCUPANCY_SWEEP
static inline unsigned occupancy_sweep_lds_pad(const char* env_name)
{
	const char* v = getenv(env_name);
	return v ? (unsigned)strtoul(v, nullptr, 10) : 0u;
}
#else
static inline constexpr unsigned occupancy_sweep_lds_pad(const char*) { return 0u; }
#endif

// Can the splat reach alpha >= 1/255 at ANY point of an axis-aligned box of pixel centres
// [bx, bx+EXT] x [by, by+EXTY]?  power(d) = -q(d), q(d) = 0.5*(a dx^2 + c dy^2) + b dx dy with
// d = centre - pixel.  For a positive-definite conic q is convex, so its minimum over the box is 0
// if the centre lies inside; otherwise it sits on an edge that FACES the centre (the level ellipse that first touches
// the box touches it at a point the centre sees).  Two clamped 1-D parabola minima cover every case without a branch
// or a select: on the line dx = dxe := clamp(0, dx_lo, dx_hi) -- the facing vertical edge if the centre is outside
// the box's x-range, the vertical line THROUGH the centre if it is inside -- and on dy = dye := clamp(0, dy_lo, dy_hi)
// likewise.  Centre inside both ranges: both lines pass through it, q = 0.  Inside the x-range only: the minimum over
// the box lies on the facing horizontal edge, which the second parabola finds; the first (a line through the box that
// ends on that edge) can only be larger.  Outside both: the two facing edges.  (Until round 5: all four edges, then
// the two facing ones chosen by compares and selects -- six of each per pair of tests, each at twice the issue cost
// of an arithmetic instruction; clamp = v_med3_f32.)  The splat is kept iff  -qmin >= power_cut - slack
// (power_cut already carries a margin; the extra slack covers the rounding of this test).  Anything not provably a
// miss -- non-PD conics, NaN conics / cuts -- is kept, so the per-pixel decisions downstream stay exact.
template <int EXT, int EXTY = EXT>   // box of pixel centres [bx, bx+EXT] x [by, by+EXTY]: EXT = 7 (quadrant) or 15 (tile)
__device__ __forceinline__ bool box_may_hit(float X, float Y, float a, float b, float c, float cut, float rb_c,
                                            float rb_a, bool pd, float bx, float by)
{
	const float dx_lo = X - (bx + (float)EXT), dx_hi = X - bx;
	const float dy_lo = Y - (by + (float)EXTY), dy_hi = Y - by;
	const float dxe = __builtin_amdgcn_fmed3f(0.0f, dx_lo, dx_hi);
	const float dy0 = __builtin_amdgcn_fmed3f(rb_c * dxe, dy_lo, dy_hi);
	const float q0 = 0.5f * (a * dxe * dxe + c * dy0 * dy0) + b * dxe * dy0;
	const float dye = __builtin_amdgcn_fmed3f(0.0f, dy_lo, dy_hi);
	const float dx1 = __builtin_amdgcn_fmed3f(rb_a * dye, dx_lo, dx_hi);
	const float q1 = 0.5f * (a * dx1 * dx1 + c * dye * dye) + b * dx1 * dye;
	const float qmin = fminf(q0, q1);
	const float slack = 1.0e-3f + 1.0e-4f * fabsf(cut);
	const bool miss = pd && (-qmin < cut - slack);
	return !miss;
}

// Staged batch of up to BATCH (<= 256) list entries + per-wave (= per 8x8 quadrant) compacted
// index lists.  The forward stages 256 entries at a time; the backward 128, which with its
// per-wave partial-sum slots keeps LDS at 25 KB per workgroup (6 workgroups per CU).
template <int BATCH>
struct TileStageT {
	float4 q0[BATCH];                // x, y, -conic a / 2, -conic b
	float4 q1[BATCH];                // -conic c / 2, power cut, opacity, depth
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
		const float a = -2.0f * r0.z, b = -r0.w, c = -2.0f * r1.x;   // the record holds (-a/2, -b, -c/2)
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

// ---- split lists: one list per 8 x (8 / NS) pixel strip of a quadrant --------------------------------------------
// With one list per quadrant a wave spends a full visit on every entry that may touch ANY of its 64 pixels, and only
// 22 of them are live on average (C3).  NS = 2 / 4 gives every 32- / 16-lane group of the wave (rows 0-3 | 4-7 of the
// quadrant, or its four 8 x 2 strips = the four DPP rows) its own list: in one visit the groups work on DIFFERENT
// entries (per-lane record addresses; a 16-lane group still reads one address, so the LDS reads stay broadcasts), and
// the wave needs max-over-groups(list length) visits instead of the quadrant's: 0.82x (NS = 2) / 0.73x (NS = 4) at C3
// (tools/walk_stats.py).  Every per-pixel operation and its order are unchanged: a lane only skips entries that the
// conservative box test proves inert for its strip, so results stay bit-identical.
// Lists are padded with the offset of a SENTINEL record (slot BATCH: no pixel can be a candidate of it) so that all
// groups run the same number of visits.
// PAD: the lists are sentinel-padded to a multiple of PAD entries (4: the forward's trips; 8: the backward's chunks)
template <int BATCH, int NS, typename LT = unsigned int, int PAD = 4>   // LT: list entry type (uint16_t where LDS is tight; BATCH << 4 must fit)
struct TileStageS {
	static constexpr int NL = 4 * NS;        // lists per tile
	static constexpr int ROW = BATCH + PAD;  // list row, sentinel-padded to the next multiple of PAD
	float4 q0[BATCH + 1];                    // x, y, -conic a / 2, -conic b    ([BATCH] = sentinel)
	float4 q1[BATCH + 1];                    // -conic c / 2, power cut, opacity, depth
	float4 q2[BATCH + 1];                    // r, g, b, -
	LT list[NL][ROW];                        // BYTE offsets (entry << 4) into q0 / q1 / q2, in list order
	unsigned int cnt[4][NL];                 // [staging wave][list]
};
#define BSR_SENTINEL_X (-1.0e15f)            // (finished forward lanes sit at +1e15: the sentinel must be far from that too)

template <int BATCH, int NS, typename LT, int PAD>
__device__ __forceinline__ void stage_init(TileStageS<BATCH, NS, LT, PAD>& st, int tid)
{
	if (tid == 0) {
		st.q0[BATCH] = make_float4(BSR_SENTINEL_X, 0.f, -0.5f, 0.f);   // power = -0.5e30: below any cut, never > 0
		st.q1[BATCH] = make_float4(-0.5f, 0.f, 0.f, 0.f);
		st.q2[BATCH] = make_float4(0.f, 0.f, 0.f, 0.f);
	}
}

// List of the calling lane: quadrant = wave, strip = its 64 / NS-lane group.
template <int NS>
__device__ __forceinline__ int my_list_index(int wave, int lane) { return wave * NS + (NS == 1 ? 0 : NS == 2 ? (lane >> 5) : (lane >> 4)); }

// As stage_and_compact, for the split lists.  Returns the number of visits the calling wave needs (the longest of
// its NS lists); st.list[l][0 .. that, rounded up to 4) is valid for each of its lists.
template <int BATCH, int NS, typename LT, int PAD>
__device__ __forceinline__ int stage_and_compact_s(TileStageS<BATCH, NS, LT, PAD>& st, int tid, bool valid, const float4 r0,
                                                   const float4 r1, const float4 r2, float tile_x0, float tile_y0,
                                                   unsigned int& hits)
{
	constexpr int NL = 4 * NS;
	constexpr int SH = 8 / NS;   // strip height in pixels
	const int wave = tid >> 6, lane = tid & 63;
	hits = 0;                    // bit l: this thread's entry may touch strip l (one VGPR, not NL flags and masks)
	if (valid) {
		st.q0[tid] = r0;
		st.q1[tid] = r1;
		st.q2[tid] = r2;
		const float a = -2.0f * r0.z, b = -r0.w, c = -2.0f * r1.x;   // the record holds (-a/2, -b, -c/2)
		const bool pd = (a > 0.0f) && (c > 0.0f) && (a * c - b * b > 0.0f);
		const float rb_c = -b / c, rb_a = -b / a;
		// one quadrant at a time (a real loop: fully unrolled, the NL tests' shared subexpressions cost 30+ VGPRs of
		// the kernel's budget), its NS strips unrolled
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)   // (vectorised it runs two quadrants in v_pk_* pairs)
		for (int q = 0; q < 4; q++) {
			const float bx = tile_x0 + (float)((q & 1) << 3), by = tile_y0 + (float)((q >> 1) << 3);
			unsigned int hq = 0;
#pragma unroll
			for (int sidx = 0; sidx < NS; sidx++)
				if (box_may_hit<7, SH - 1>(r0.x, r0.y, a, b, c, r1.y, rb_c, rb_a, pd, bx, by + (float)(sidx * SH))) hq |= 1u << sidx;
			hits |= hq << (q * NS);
		}
	}
#pragma unroll
	for (int l = 0; l < NL; l++) {
		const unsigned long long m = wave_ballot((hits >> l) & 1u);
		if (lane == 0) st.cnt[wave][l] = (unsigned int)__popcll(m);
	}
	__syncthreads();
	const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
	for (int l = 0; l < NL; l++) {
		const unsigned long long m = wave_ballot((hits >> l) & 1u);
		if ((hits >> l) & 1u) {
			const unsigned int off = (wave > 0 ? st.cnt[0][l] : 0u) + (wave > 1 ? st.cnt[1][l] : 0u) +
			                         (wave > 2 ? st.cnt[2][l] : 0u);
			st.list[l][off + (unsigned int)__popcll(m & lt)] = (LT)(tid << 4);
		}
	}
	// each wave pads ITS lists with the sentinel up to the longest of them (rounded up to the four entries the walk
	// reads per trip); nobody else writes there
	int tot[NS];
	int longest = 0;
#pragma unroll
	for (int sidx = 0; sidx < NS; sidx++) {
		const int l = wave * NS + sidx;
		tot[sidx] = (int)(st.cnt[0][l] + st.cnt[1][l] + st.cnt[2][l] + st.cnt[3][l]);
		longest = max(longest, tot[sidx]);
	}
	const int padded = (longest + (PAD - 1)) & ~(PAD - 1);
#pragma unroll
	for (int sidx = 0; sidx < NS; sidx++)
		for (int i = tot[sidx] + lane; i < padded; i += 64) st.list[wave * NS + sidx][i] = (LT)(BATCH << 4);
	__syncthreads();
	return longest;
}
template <int BATCH> __device__ __forceinline__ float4 srec_q0(const char* r) { return *reinterpret_cast<const float4*>(r); }
template <int BATCH> __device__ __forceinline__ float4 srec_q1(const char* r) { return *reinterpret_cast<const float4*>(r + (BATCH + 1) * 16); }
template <int BATCH> __device__ __forceinline__ float2 srec_q1lo(const char* r) { return *reinterpret_cast<const float2*>(r + (BATCH + 1) * 16); }
template <int BATCH> __device__ __forceinline__ float2 srec_q1hi(const char* r) { return *reinterpret_cast<const float2*>(r + (BATCH + 1) * 16 + 8); }
template <int BATCH> __device__ __forceinline__ float4 srec_q2(const char* r) { return *reinterpret_cast<const float4*>(r + (BATCH + 1) * 32); }

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
