// Forward tile renderer: front-to-back alpha blending of colour + depth per 16x16 tile.
//
// Semantics: reference renderCUDA, cuda_rasterizer/forward.cu:341-471 (under
// /root/reference/submodules/depth-diff-gaussian-rasterization).  Per pixel the arithmetic and
// every decision (power > 0, alpha < 1/255, T*(1-alpha) < 1e-4, depth normalisation) are
// evaluated in the reference's order.  EXACT = true (BSR_FLAG_EXACT_EXP of bsr_forward_ex): exp(power) is the pinned
// bsr_expf on every evaluation and the results match the CPU oracle bit for bit.  EXACT = false (the default):
// the VALUE of exp(power) comes from v_exp_f32 (1 ulp, 2 issue slots instead of 13) and only a wave with a pixel
// inside the decision band around the alpha >= 1/255 cut evaluates the pinned exp -- see the visit below.
//
// MI355X mapping: one tile = one 256-thread workgroup = 4 wave64; each wave owns an 8x8 pixel
// quadrant.  List entries are staged 256 at a time: each thread gathers one 48-byte splat record
// (3 x 16-B loads out of its one 64-B line in the L2/Infinity-Cache resident record table), runs a conservative
// ellipse-vs-strip test on it and the batch is compacted with wave ballots + prefix popcounts
// into ordered index lists, one per 8 x 4 pixel half of each quadrant (NS = 2, tile_common.h "split lists"): the two
// 32-lane halves of a wave walk their own lists side by side (0.82x the visits of one list per quadrant at C3;
// k_render_fwd -6..8 % at C3 / C5, unchanged on dense and on sparse views; one list per 8 x 2 strip (NS = 4) halves
// the visits' gain again but its 16 box tests per entry and 29 KB of LDS cost more than it saves).  The per-pixel reject
// path needs 24 B of LDS
// and no exp: `power < power_cut` (precomputed -ln(255*opacity) minus a margin) proves
// alpha < 1/255 without evaluating it.  The inner loop is wave-uniform (one ballot per entry).
#include "tile_common.h"

namespace bsr {

// ---- diagnostic build only (make stats; tools/walk_stats.py --forward): counters of the forward walk + a timeline
#ifdef BSR_WALK_STATS
#define BSR_NSTAT_F 24
__device__ unsigned long long g_fwd_stats[BSR_NSTAT_F];
#define FSTAT_ADD(i, v) (wstat[i] += (unsigned long long)(v))
#else
#define FSTAT_ADD(i, v) ((void)0)
#endif
// (make timeline: the per-workgroup start / end stamps alone -- two clock reads and one 32-byte store per workgroup, the
// walk itself as shipped; the counters above slow the kernel ~10x and distort the stamps)
#if defined(BSR_WALK_STATS) || defined(BSR_WALK_TIMELINE)
__device__ unsigned long long g_fwd_times[4 * 70000];
#endif

// View-batched calls (bsr_forward_views) stack their views into one virtual image of n_views * gy tile rows: tile
// row tyv belongs to view tyv / gy; pixel coordinates, image outputs and the per-pixel state are per view.
#ifndef BSR_FWD_BATCH
#define BSR_FWD_BATCH BSR_BLOCK   // entries staged per batch (A/B hook)
#endif
#ifdef BSR_FWD_WAVES              // A/B hook: occupancy target handed to the register allocator
#define BSR_FWD_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(BSR_FWD_WAVES, BSR_FWD_WAVES)))
#else
#define BSR_FWD_WAVES_ATTR
#endif
template <int NS, int FB, bool EXACT>
__global__ void __launch_bounds__(BSR_BLOCK) BSR_FWD_WAVES_ATTR k_render_fwd(int n_tiles, int gx, int gy, int W, int H,
                                                          const int* __restrict__ n_ptr, int capacity, int nan_on_overflow,
                                                          const uint2* __restrict__ tile_range,
                                                          uint32_t* point_list, int* __restrict__ masks_flag,
                                                          const float4* __restrict__ rec,
                                                          const float* __restrict__ bg_color,
                                                          float* __restrict__ final_T,
                                                          uint32_t* __restrict__ n_contrib,
                                                          float* __restrict__ out_color,
                                                          float* __restrict__ out_depth, int* __restrict__ pool_ctr)
{
	__shared__ TileStageS<FB, NS> st;
	__shared__ int s_done[4];
	__shared__ int s_slot;

	const int tile = pooled_tile(blockIdx.x, n_tiles, pool_ctr, &s_slot);
	if (tile < 0) return;
	// the three scalar loads leave together (a sparse view's tile lives for little more than its chain of dependent loads)
	const int n_instances = *n_ptr;
	const uint2 range = tile_range[tile];
	const uint32_t start = range.x, end = range.y;
	if (n_instances > capacity) {
		// launched ahead of the host's read-back with too small a scratch: the default path re-runs the tail.  A
		// BSR_FLAG_NO_READBACK call has no re-run: its frame says so itself (NaN), never a stale or half-written image
		if (nan_on_overflow) {
			const int tx = tile % gx, ty = tile / gx;   // (single-view calls only)
			const int px = tx * BSR_TILE + (threadIdx.x & 15), py = ty * BSR_TILE + (threadIdx.x >> 4);
			if (px < W && py < H) {
				const size_t plane = (size_t)H * W, pix_id = (size_t)W * py + px;
				const float nan = __builtin_nanf("");
				out_color[pix_id] = nan;
				out_color[plane + pix_id] = nan;
				out_color[2 * plane + pix_id] = nan;
				out_depth[pix_id] = nan;
				// the saved image state too (GaussianRasterizer(return_alpha=True) returns 1 - final_T): NaN, never
				// whatever the uninitialised scratch held
				if (final_T != nullptr) final_T[pix_id] = nan;
				if (n_contrib != nullptr) n_contrib[pix_id] = 0u;
			}
		}
		return;
	}
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (wave-uniform: kept on the scalar side)
	// Hand-over to the backward walk (masks_flag != nullptr: one view, ids below 2^24): the NS = 2 staging below has the
	// eight per-half box tests of every entry it stages in one byte; written into the top byte of the entry's point_list
	// word, the backward's waves take their lists from it instead of gathering and testing the records again (the tests
	// are a pure function of the sorted list: built once, here).  *masks_flag says whether THIS run did so.
	const bool hand_over = (NS == 2) && masks_flag != nullptr;
	if (masks_flag != nullptr && tile == 0 && tid == 0) *masks_flag = hand_over ? 1 : 0;
#ifdef BSR_WALK_STATS
	unsigned long long wstat[BSR_NSTAT_F] = {};
#endif
#if defined(BSR_WALK_STATS) || defined(BSR_WALK_TIMELINE)
	const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
	// the sentinel record, written by EVERY wave (the same three values): a wave that stages a small tile for itself
	// (below) reads it behind its own wave-level fence, the workgroup-staged path behind the staging's barriers
	stage_init(st, lane);
	const unsigned int* const my_list = &st.list[my_list_index<NS>(wave, lane)][0];
	const int tx = tile % gx, tyv = tile / gx;
	const int view = tyv / gy, ty = tyv - view * gy;
	const int px = tx * BSR_TILE + ((wave & 1) << 3) + (lane & 7);
	const int py = ty * BSR_TILE + ((wave >> 1) << 3) + (lane >> 3);
	const bool inside = px < W && py < H;
	// A finished lane (outside the image, or stopped at T < 1e-4) gets its pixel centre moved ~1e15 px
	// away: `power` then falls far below any cut, so the per-entry candidate vote needs no `!done` term.
	// (Only an optimisation of the vote: the blend predicate below still carries `!done`.)
	float pixfx = inside ? (float)px : 1.0e15f;
	const float pixfy = (float)py;
	const float tile_x0 = (float)(tx * BSR_TILE), tile_y0 = (float)(ty * BSR_TILE);

	const int n = (int)(end - start);

	bool done = !inside;
	float T = 1.0f;
	uint32_t last16 = 0;   // 16 * last_contributor (list offsets are kept pre-scaled)
	float C0 = 0.f, C1 = 0.f, C2 = 0.f;
	float D = 0.f;
	float acc = 0.000001f;

	// the walk over the calling wave's list (n_mine entries staged; pos_bias: 16 * (list position + 1) - byte offset of
	// the staged slot), shared by the two staging schemes below
	auto walk = [&](const int n_mine, const uint32_t pos_bias) {
		// Walk over this wave's compacted list(s), FOUR entries per trip as one straight-line block: no vote and
		// no branch per entry.  (The predecessor voted "any candidate lane?" per entry and returned early: 2 % of the
		// visits took that exit, and every visit paid two dependent LDS round trips and three branches in a chain
		// the seven waves of a SIMD could not cover -- once v_exp_f32 had cut the arithmetic, a quarter fewer
		// instructions bought 4 %.)  Here the four entries' LDS reads and their power / exp / alpha chains are
		// independent and the scheduler interleaves them; only the blend itself is serial in T.  The rows are
		// sentinel-padded to a multiple of four, and a sentinel is never a candidate.
		// Fully predicated by value: a lane that must not blend carries alpha 0.
		//   not a candidate, alpha < 1/255, or done -> a_eff = 0   (reference: continue)
		//   T (1 - alpha) < 1e-4      -> a       = 0   and the lane is done (reference :433-437)
		// T >= 1e-4 is an invariant of every lane, so test_T < 1e-4 can only fire where a_eff > 0.
		int n_lim = __builtin_amdgcn_readfirstlane(n_mine);   // set to 0 to leave (single loop exit)
		uint32_t lastj = 0xffffffffu;
		const char* const rec0 = reinterpret_cast<const char*>(&st.q0[0]);
		struct Ent { float4 q0, q1, q2; float power; bool cand; };
		auto load = [&](const unsigned int joff) {
			Ent e;
			const char* rec = rec0 + joff;
			e.q0 = srec_q0<FB>(rec);      // x, y, -a/2, -b
			e.q1 = srec_q1<FB>(rec);      // -c/2, power cut, opacity, depth
			e.q2 = srec_q2<FB>(rec);      // r, g, b, centre of the decision band
			const float dx = e.q0.x - pixfx;
			const float dy = e.q0.y - pixfy;
			e.power = (e.q0.z * dx * dx + e.q1.x * dy * dy) + e.q0.w * dx * dy;   // pre-scaled conic: == -0.5f * (a dx dx + c dy dy) - b dx dy
			// reference: if (power > 0) continue;  then alpha < 1/255 -> continue (here proven by the cut)
			e.cand = !(e.power > 0.0f) && !(e.power < e.q1.y);
			return e;
		};
		// c2: this lane blends the entry unless the stop test below fires
		auto blend = [&](const Ent& e, const float alpha_raw, const bool c2, const unsigned int joff) {
			const float a_eff = c2 ? alpha_raw : 0.0f;
			bool stop;
			if (EXACT) {
				const float test_T = T * (1 - a_eff);
				stop = test_T < 0.0001f;
				const float a = stop ? 0.0f : a_eff;
				// reference :439-446 as its nvcc build evaluates them (and the oracle restates them): c * alpha rounded,
				// then ONE fused multiply-add with T.  fma(c * 0, T, x) == x exactly for every finite c.
				C0 = __builtin_fmaf(e.q2.x * a, T, C0);
				C1 = __builtin_fmaf(e.q2.y * a, T, C1);
				C2 = __builtin_fmaf(e.q2.z * a, T, C2);
				D = __builtin_fmaf(e.q1.w * a, T, D);
				acc = __builtin_fmaf(a, T, acc);
				T = stop ? T : test_T;
			} else {
				// default mode (values within ulps, not bit-equal: DESIGN.md "Numerics"): the blend weight alpha T once, then
				// one fused multiply-add per target -- 6 instead of 9 instructions; (c alpha) T and c (alpha T) differ by
				// one rounding.  T itself keeps the reference's form T (1 - alpha): 1 - alpha is exact for alpha near 1, so
				// the product carries half an ulp; T - alpha T would cancel (50 ulp of T' at the 0.99 clamp) and the backward,
				// which rebuilds T by dividing by the same (1 - alpha), would see it (measured: dL_dmeans3D 1.1e-5 of scale
				// at C3 instead of 3e-7)
				const float test_T = T * (1 - a_eff);
				stop = test_T < 0.0001f;
				const float w = (stop ? 0.0f : a_eff) * T;
				C0 = __builtin_fmaf(e.q2.x, w, C0);
				C1 = __builtin_fmaf(e.q2.y, w, C1);
				C2 = __builtin_fmaf(e.q2.z, w, C2);
				D = __builtin_fmaf(e.q1.w, w, D);
				acc = acc + w;
				T = stop ? T : test_T;
			}
			lastj = (c2 && !stop) ? joff : lastj;   // byte offset of the last entry of THIS batch the lane blended
			done = done || stop;
#ifdef BSR_WALK_STATS
			{
				const int live = __popcll(wave_ballot(c2 && !stop));
				FSTAT_ADD(0, 1);                                        // visits (incl. sentinel padding)
				FSTAT_ADD(1, wave_ballot(e.cand) != 0ull ? 1 : 0);      // ... with a candidate lane
				FSTAT_ADD(5, live);                                     // lanes that blend
				if (live) { FSTAT_ADD(3, 1); FSTAT_ADD(8 + ((live - 1) >> 3), 1); }   // blending visits + histogram
			}
#endif
		};
		for (int i = 0; i < n_lim; i += 4) {
			const uint4 l = *reinterpret_cast<const uint4*>(my_list + i);   // (the row is sentinel-padded to a multiple of 4)
			const Ent e0 = load(l.x), e1 = load(l.y), e2 = load(l.z), e3 = load(l.w);
			float g0, g1, g2, g3;
			bool decide = true;   // false: alpha >= 1/255 is already proven for every candidate lane of the trip
			if (EXACT) {
				g0 = bsr_expf_walk(e0.power);
				g1 = bsr_expf_walk(e1.power);
				g2 = bsr_expf_walk(e2.power);
				g3 = bsr_expf_walk(e3.power);
			} else {
				// `alpha >= 1/255` is decided on alpha = min(0.99, o E(power)) with the pinned exp E.  A candidate has
				// power >= power_cut = -ln(255 o) - 1e-3; at power >= -ln(255 o) + 1.1e-3 ANY exp within a few ulp gives
				// alpha >= (1 + 1e-3) / 255: the decision is "blend" whatever the last bits are (margin 1e-3 in the
				// exponent = 0.1 % of alpha, against 1e-6 for the rounding of logf, of E and of v_exp_f32 together).
				// Only a lane inside that 2.1e-3 wide band (centre = power_cut + 1e-3, staged in q2.w) needs E to decide, and
				// only a trip holding such a lane evaluates E at all (a few % of the trips); every other one takes the
				// VALUE from v_exp_f32.  Which exp a lane uses depends on ITS power alone -- never on the lanes or
				// entries it shares a trip with -- so the image is the same whichever instantiation (NS, view batching,
				// scratch capacity) renders it.  The test is `!(|power - centre| >= 1.1e-3)` so that a NaN (centre:
				// opacity <= 0 or NaN; power: NaN conic) counts as inside -- those lanes get the exact path's arithmetic,
				// e.g. alpha < 0 is skipped as in the reference.  (One subtraction and ONE compare per entry, voted
				// directly: a vote on `cand && power < hi` makes hipcc materialise the AND-ed mask in a VGPR and compare
				// it again -- two more half-rate VALU slots per entry, 8 % of the trip.)  The backward selects the exp
				// per lane with the same expression (render_bwd.hip) and so sees the same alpha.
				const bool b0 = !(fabsf(e0.power - e0.q2.w) >= 1.1e-3f), b1 = !(fabsf(e1.power - e1.q2.w) >= 1.1e-3f);
				const bool b2 = !(fabsf(e2.power - e2.q2.w) >= 1.1e-3f), b3 = !(fabsf(e3.power - e3.q2.w) >= 1.1e-3f);
				g0 = __builtin_amdgcn_exp2f(e0.power * 1.44269504088896341f);
				g1 = __builtin_amdgcn_exp2f(e1.power * 1.44269504088896341f);
				g2 = __builtin_amdgcn_exp2f(e2.power * 1.44269504088896341f);
				g3 = __builtin_amdgcn_exp2f(e3.power * 1.44269504088896341f);
				decide = (wave_ballot(b0) | wave_ballot(b1) | wave_ballot(b2) | wave_ballot(b3)) != 0ull;   // rare
				if (decide) {
					FSTAT_ADD(2, 1);   // trips decided by the pinned exp
					g0 = b0 ? bsr_expf_walk(e0.power) : g0;
					g1 = b1 ? bsr_expf_walk(e1.power) : g1;
					g2 = b2 ? bsr_expf_walk(e2.power) : g2;
					g3 = b3 ? bsr_expf_walk(e3.power) : g3;
				}
			}
			const float a0 = fminf(0.99f, e0.q1.z * g0), a1 = fminf(0.99f, e1.q1.z * g1);
			const float a2 = fminf(0.99f, e2.q1.z * g2), a3 = fminf(0.99f, e3.q1.z * g3);
			if (decide) {
				blend(e0, a0, e0.cand && !(a0 < 1.0f / 255.0f) && !done, l.x);
				blend(e1, a1, e1.cand && !(a1 < 1.0f / 255.0f) && !done, l.y);
				blend(e2, a2, e2.cand && !(a2 < 1.0f / 255.0f) && !done, l.z);
				blend(e3, a3, e3.cand && !(a3 < 1.0f / 255.0f) && !done, l.w);
			} else {
				blend(e0, a0, e0.cand && !done, l.x);
				blend(e1, a1, e1.cand && !done, l.y);
				blend(e2, a2, e2.cand && !done, l.z);
				blend(e3, a3, e3.cand && !done, l.w);
			}
			if (wave_ballot(!done) == 0ull) n_lim = 0;   // every pixel of the quadrant is done: leave
		}
		last16 = lastj != 0xffffffffu ? lastj + pos_bias : last16;   // 16 * (list position + 1)
	};

	// A tile of at most 64 entries (the views of a camera sweep: a few dozen entries in every tile) is staged by each
	// wave FOR ITSELF: the wave loads the entries, tests them against its own quadrant, and keeps records, list and a
	// copy of the sentinel in its own quarter of the staging arrays -- no workgroup barrier anywhere, the four waves
	// run their chains of dependent loads independently (such a tile's life is those chains, not arithmetic).  Same
	// box test, same list order, same walk: identical bits.
	if ((NS == 1) && n <= 64) {
		// (outside the batch loop: its address arithmetic must not live across the other path's walk)
		if (n > 0 && wave_ballot(!done) != 0ull) {
			int n_mine;
			uint32_t pos_bias;
			const bool valid = lane < n;
			bool hit = false;
			if (valid) {
				const uint32_t id = point_list[start + lane];
				const float4* r = rec + (size_t)id * BSR_REC;
				const float4 r0 = r[0], r1 = r[1];
				float4 r2 = r[2];
				r2.w = r1.y + 1.0e-3f;
				const int slot = (wave << 6) + lane;
				st.q0[slot] = r0;
				st.q1[slot] = r1;
				st.q2[slot] = r2;
				const float a = -2.0f * r0.z, b = -r0.w, c = -2.0f * r1.x;   // the record holds (-a/2, -b, -c/2)
				const bool pd = (a > 0.0f) && (c > 0.0f) && (a * c - b * b > 0.0f);
				hit = box_may_hit<7, 7>(r0.x, r0.y, a, b, c, r1.y, -b / c, -b / a, pd, tile_x0 + (float)((wave & 1) << 3),
				                        tile_y0 + (float)((wave >> 1) << 3));
			}
			const unsigned long long m = wave_ballot(hit);
			n_mine = (int)__popcll(m);
			if (hit) st.list[wave][__popcll(m & ((1ull << lane) - 1ull))] = (unsigned int)(((wave << 6) + lane) << 4);
			if (lane < 3) st.list[wave][n_mine + lane] = (unsigned int)(FB << 4);   // pad to the four entries of a trip
			pos_bias = 16u - ((uint32_t)wave << 10);
			FSTAT_ADD(6, 1);    // batches (per wave)
			FSTAT_ADD(7, n);    // staged entries
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			walk(n_mine, pos_bias);
		}
	} else {
		for (int base = 0; base < n; base += FB) {
			const bool wave_done = (wave_ballot(!done) == 0ull);
			int n_mine;
			uint32_t pos_bias;
			if (base > 0) {
				// whole tile finished?  (the barrier is also the WAR fence for the staging buffers; the first batch has
				// nothing to wait for: the sentinel stage_init wrote is read behind the two barriers of the staging)
				if (lane == 0) s_done[wave] = wave_done ? 1 : 0;
				__syncthreads();
				if (s_done[0] & s_done[1] & s_done[2] & s_done[3]) break;
			}
			const int cnt = min(FB, n - base);
			const bool valid = tid < cnt;
			float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
			uint32_t id = 0;
			if (valid) {
				id = point_list[start + base + tid];
				const float4* r = rec + (size_t)id * BSR_REC;
				r0 = r[0];
				r1 = r[1];
				r2 = r[2];
				r2.w = r1.y + 1.0e-3f;   // staged q2.w (in HBM: half of the kept-tile mask, backward only): centre of the decision band, -ln(255 o)
			}
			unsigned int hits;   // bit 2 q + h: the entry may touch rows 4 h .. 4 h + 3 of quadrant q
			n_mine = stage_and_compact_s(st, tid, valid, r0, r1, r2, tile_x0, tile_y0, hits);
			if (hand_over && valid) point_list[start + base + tid] = id | (hits << 24);
			pos_bias = (uint32_t)(base + 1) << 4;
			FSTAT_ADD(6, 1);      // batches (per wave)
			FSTAT_ADD(7, cnt);    // staged entries (every wave sees the batch)
			if (!wave_done) walk(n_mine, pos_bias);
		}
	}

	if (inside) {
		const size_t plane = (size_t)H * W;
		const size_t pix_id = (size_t)W * py + px, img = (size_t)view * plane;
		if (final_T != nullptr) {   // (view-batched calls have no backward and pass no per-pixel state)
			final_T[img + pix_id] = T;
			n_contrib[img + pix_id] = last16 >> 4;
		}
		float* oc = out_color + 3 * img;
		oc[pix_id] = __builtin_fmaf(T, bg_color[0], C0);
		oc[plane + pix_id] = __builtin_fmaf(T, bg_color[1], C1);
		oc[2 * plane + pix_id] = __builtin_fmaf(T, bg_color[2], C2);
		out_depth[img + pix_id] = (acc > 0.5f) ? D / acc : 0.0f;
	}
#ifdef BSR_WALK_STATS
	if (lane == 0)
		for (int i = 0; i < BSR_NSTAT_F; i++)
			if (wstat[i]) atomicAdd(&g_fwd_stats[i], wstat[i]);
#endif
#if defined(BSR_WALK_STATS) || defined(BSR_WALK_TIMELINE)
	if (lane == 0 && wave == 0 && tile < 70000) {   // (filed under the TILE: every tile is rendered exactly once per launch)
		unsigned long long* t = g_fwd_times + 4 * (size_t)tile;
		uint32_t xcc, hwid;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
		t[0] = t_start;
		t[1] = __builtin_amdgcn_s_memrealtime();
		t[2] = (unsigned long long)(xcc & 0xfu) | ((unsigned long long)blockIdx.x << 4) | ((unsigned long long)hwid << 32);
		t[3] = (unsigned long long)tile | ((unsigned long long)n << 32);
	}
#endif
}

void launch_render_fwd(int gx, int gy, int n_views, int W, int H, const int* n_ptr, int capacity, const uint2* tile_range,
                       uint32_t* point_list, int* masks_flag,
                       const float4* rec, const float* bg, float* final_T, uint32_t* n_contrib, float* out_color,
                       float* out_depth, bool exact, bool nan_on_overflow, int* pool_ctr, hipStream_t s)
{
	const int n_tiles = gx * gy * n_views;
	const int blocks = pooled_grid(n_tiles);
	// split lists pay off once tiles hold a few dozen entries (C3: 360, C5: 1800); on the sparse views of a camera sweep
	// (a dozen entries per tile, most tiles empty) their 8 box tests per entry and the padding are pure overhead.
	// Both instantiations produce identical bits.
	const unsigned pad = occupancy_sweep_lds_pad("BSR_SWEEP_LDS_PAD_FWD");
	const bool split = (long long)capacity >= 48ll * n_tiles;
#define BSR_LAUNCH_FWD(NS_, EX_)                                                                                        \
	hipLaunchKernelGGL((k_render_fwd<NS_, BSR_FWD_BATCH, EX_>), dim3(blocks), dim3(BSR_BLOCK), pad, s, n_tiles, gx, gy, W, H, \
	                   n_ptr, capacity, nan_on_overflow ? 1 : 0, tile_range, point_list, masks_flag, rec, bg, final_T, n_contrib, out_color, out_depth, pool_ctr)
	if (split && exact) BSR_LAUNCH_FWD(2, true);
	else if (split) BSR_LAUNCH_FWD(2, false);
	else if (exact) BSR_LAUNCH_FWD(1, true);
	else BSR_LAUNCH_FWD(1, false);
#undef BSR_LAUNCH_FWD
}

}  // namespace bsr

#if defined(BSR_WALK_STATS) || defined(BSR_WALK_TIMELINE)
// Diagnostic builds only: copies (and clears) the forward counters (which = 2; stats build) / its timeline (3).
extern "C" int bsr_debug_walk_stats_fwd(int which, void* out, size_t bytes)
{
	hipError_t e = hipDeviceSynchronize();
#ifdef BSR_WALK_STATS
	if (e == hipSuccess && which == 2) {
		e = hipMemcpyFromSymbol(out, HIP_SYMBOL(bsr::g_fwd_stats), bytes);
		static unsigned long long zeros[BSR_NSTAT_F];
		if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(bsr::g_fwd_stats), zeros, sizeof(zeros));
		return e == hipSuccess ? 0 : 1;
	}
#endif
	if (which != 3) return 1;
	if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(bsr::g_fwd_times), bytes);
	return e == hipSuccess ? 0 : 1;
}
#endif
