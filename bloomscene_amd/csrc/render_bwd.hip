// Backward tile renderer: back-to-front re-traversal producing, per list entry (= per
// (tile, Gaussian) instance), the nine partial sums dL/d(mean2D.xy, conic.xyw, opacity, colour.rgb).
//
// Semantics: reference renderCUDA (backward), cuda_rasterizer/backward.cu:399-586 (under
// /root/reference/submodules/depth-diff-gaussian-rasterization); dL_depths is ignored exactly as
// there (:457-463,539-554 are commented out).  Per (pixel, Gaussian) pair the nine contributions
// are computed with the reference's operation order.
//
// The reference issues 9 lane-scattered float atomicAdds per pair.  On MI355X lane-scattered
// global float atomics run ~17x below the contiguous rate (~20 G/s chip-wide), which would cap this
// kernel at ~2 ms for C3.  Instead NO global atomic is issued at all:
//   * the 9 partials are summed over the wave's 64 pixels by a halving reduction on lane-masked DPP
//     writes (wave_sums_masked below),
//   * each wave stores its sums in its own LDS slot (LDS float atomics cost ~16 cycles each on
//     gfx950 whatever the exec mask); the 4 slots are added in a fixed order at the batch end,
//   * each instance's 9 sums are written ONCE to its row [9 floats; 10 with the depth gradient] of a GAUSSIAN-MAJOR slab: the
//     rows of one Gaussian (one per kept tile of its rect, row-major) are adjacent, at the instance
//     numbering fixed by the forward's preprocess,
//   * k_preprocess_bwd later reads each Gaussian's rows as one contiguous run and adds them in that
//     fixed order.  (An earlier version wrote rows in list order and gathered them through an
//     instance->slot map: 48-B rows fetched at random cost 1.75 sectors each and the map another
//     scattered pass; moving the scatter to the write side made k_preprocess_bwd 23 % faster.)
// As in the forward pass, a staged batch is first compacted per 8x8 quadrant (tile_common.h).
#include "tile_common.h"

namespace bsr {

// ---- halving reduction of 9 (10) values over the 64 lanes: lane-masked DPP writes ------------
// Nine independent 6-step reductions would be 54 cross-lane adds.  Instead the values are split
// between partner lanes at every step, halving the live set; a first version selected the kept value
// per lane (2 selects + 1 DPP add per pair, 33 instructions).  Measured on MI355X
// (tools/microbench/valu_rates.hip, 8 waves/SIMD): v_fma/v_mul issue every ~2.6 cycles per SIMD, but
// v_cndmask (SGPR mask), v_cmp -> SGPR and every DPP add every ~4.3.  So the halving steps run over the lane
// bits whose DPP writes the hardware can mask -- bit 2 and 3 through bank_mask (banks of 4 lanes),
// bit 4 and 5 through v_permlane16/32_swap of a PAIR (swap, then one add) -- so a pair-step costs 2
// instructions and no select; the plain steps over bits 0 and 1 come last, on the single survivor.
// 24 instructions for 9 values (25 for 10) instead of 33 (35), in place in the input registers.
// Which lane ends with which component is not assumed: calibrate_components() runs the reduction once
// on constants and reads the mapping off the result.
// All ten values are declared in/out so that no two of them can be given the same register (two inputs
// holding the same SSA value otherwise could, and the block overwrites x0..x3, x8 in place).
// Hazards: inline asm gets no automatic wait states; a DPP/permlane read needs 2 after a VALU write
// of the same VGPR -- the order below keeps >= 2 instructions between, s_nop where it cannot.
template <bool TEN>
__device__ __forceinline__ float wave_sums_masked(float x0, float x1, float x2, float x3, float x4, float x5, float x6,
                                                  float x7, float x8, float x9)
{
	float t;
	if (TEN) {
		asm volatile(
		    "s_nop 1\n"
		    "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %0, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %1, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %2, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %3, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %4, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n"
		    "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n"
		    "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n"
		    "v_add_f32_dpp %1, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n"
		    "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n"
		    "v_mov_b32 %5, %4\n"
		    "v_permlane16_swap_b32 %0, %1\n"
		    "v_add_f32 %0, %0, %1\n"
		    "v_permlane16_swap_b32 %4, %5\n"
		    "v_add_f32 %4, %4, %5\n"
		    "s_nop 1\n"
		    "v_permlane32_swap_b32 %0, %4\n"
		    "v_add_f32 %0, %0, %4\n"
		    "s_nop 1\n"
		    "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "s_nop 1\n"
		    "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x8), "=&v"(t), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x9));
	} else {
		asm volatile(
		    "s_nop 1\n"
		    "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %0, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %1, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %2, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %3, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n"
		    "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n"
		    "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n"
		    "v_add_f32_dpp %1, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n"
		    "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n"
		    "v_mov_b32 %5, %4\n"
		    "v_permlane16_swap_b32 %0, %1\n"
		    "v_add_f32 %0, %0, %1\n"
		    "v_permlane16_swap_b32 %4, %5\n"
		    "v_add_f32 %4, %4, %5\n"
		    "s_nop 1\n"
		    "v_permlane32_swap_b32 %0, %4\n"
		    "v_add_f32 %0, %0, %4\n"
		    "s_nop 1\n"
		    "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "s_nop 1\n"
		    "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x8), "=&v"(t), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
	}
	(void)t;
	return x0;
}

// Component (0..NV-1) whose wave total this lane holds after wave_sums_masked, and whether this lane is
// the one that stores it (the lowest lane holding that component).  Sums of small integers are exact.
template <bool TEN>
__device__ __forceinline__ int calibrate_components(int lane, bool& stores)
{
	const float r = wave_sums_masked<TEN>(0.f, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f, 9.f);
	const int comp = (int)(r * (1.0f / 64.0f));
	stores = false;
#pragma unroll
	for (int c = 0; c < (TEN ? 10 : 9); c++) {
		const uint64_t m = wave_ballot(comp == c);
		if (comp == c) stores = (m != 0ull) && (lane == (int)__builtin_ctzll(m));
	}
	return comp;
}

// Index of this tile's instance of a Gaussian in the Gaussian-major order of KEPT instances (its
// block starts at inst_offset and enumerates the kept tiles of its rect row-major), from q3 / q2.w of
// the splat record.
__device__ __forceinline__ uint32_t instance_index(const uint32_t* __restrict__ wg_base, uint32_t id, const float4 q2,
                                                   const float4 q3, int tx, int ty)
{
	const uint32_t off = wg_base[id >> 8] + __float_as_uint(q3.x), lo = __float_as_uint(q3.y), wh = __float_as_uint(q3.z);
	const uint32_t xmin = lo & 0xffffu, ymin = lo >> 16, w = wh & 0xffffu, h = wh >> 16;
	const uint64_t mask = ((uint64_t)__float_as_uint(q2.w) << 32) | (uint64_t)__float_as_uint(q3.w);
	const uint32_t k = ((uint32_t)ty - ymin) * w + ((uint32_t)tx - xmin);
	return off + kept_rank(w * h, mask, k);
}

// ---- k_render_bwd_strict: BSR_FLAG_EXACT_GRAD ----------------------------------------------------------------------
// The reference's per-pair operations on the reference's operands (backward.cu:521,527-536,557,561-583): IEEE
// divisions, no fp contraction, the pinned exp on every pair, accum_rec channel by channel, the per-pair terms summed as
// they are -- only the ORDER of the nine sums then differs from the oracle's.  The nine values of a visit are summed over
// the wave by the masked DPP network above and filed per wave; 128-entry batches staged by waves 0 and 1
// (stage_and_compact).  Until round 4 this walk (with the arithmetic shortcuts of k_render_bwd_t switched on one by one
// through attribution builds: docs/EXPERIMENTS.md) was also the default; ~2x the time of k_render_bwd_t.
#define BSR_BWD_BATCH 128
// Row stride of the per-wave partial sums: the 9 (10) storing lanes of one entry write part[wave][k][j] for
// k = 0..NV-1 with ONE ds_write_b32 (bank = dword address mod 32, lanes of a 32-lane half conflict).  With rows of
// 128 floats all of them hit one bank; 129 puts component k on bank (k + j) mod 32.
#define BSR_BWD_ROW (BSR_BWD_BATCH + 1)
template <int NV>
struct BwdShared {
	TileStageT<BSR_BWD_BATCH> st;
	float part[4][NV][BSR_BWD_ROW];   // per-wave partial sums of the current batch (plain stores)
	uint32_t max_contrib[4];
};

// State a pixel carries along the list (back to front) and the constants of the pixel.
struct PairState {
	float T;
	float acc_rec[4], last_color[4], last_alpha;   // reference :527-536 channel by channel ([3] = depth, extension)
};
struct PixelConst {
	float dpx0, dpx1, dpx2, neg_Tfinal_bg, gz, g1, ddelx_dx, ddely_dy;
};
// Per pair the reference adds (:574-583), with dL_dG = o * dL_dalpha:
//   dL_dmean2D.x += dL_dG * dG_ddelx * ddelx_dx      dL_dconic.x += -0.5 gdx dx dL_dG
//   dL_dmean2D.y += dL_dG * dG_ddely * ddely_dy      dL_dconic.y += -0.5 gdx dy dL_dG
//   dL_dopacity  += G * dL_dalpha                    dL_dconic.w += -0.5 gdy dy dL_dG
// (source order: this translation unit is built with -ffp-contract=off)
template <bool DEPTH>
__device__ __forceinline__ void pair_terms_reference(PairState& st, const PixelConst& px, const float4 q0, const float4 q1,
                                                     const float4 q2, const float dx, const float dy, const float G,
                                                     const float alpha, float (&v)[10])
{
	const float om = 1.f - alpha;
	const float inv = 1.0f / om;
	st.T = st.T / om;   // reference :521
	const float T = st.T;
	// accum_rec = last_alpha * last_color + (1 - last_alpha) * accum_rec, then (c - accum_rec) * dL_dpixel.  A pair the
	// reference skips must leave (accum_rec, last_color, last_alpha) standing.  The depth extension is the oracle's
	// separate pass (bsro_render_backward_depth): its own recurrence on d_i = gz z_i + g1.
	const bool live = G != 0.f;
	const float c4[4] = {q2.x, q2.y, q2.z, DEPTH ? px.gz * q1.w + px.g1 : 0.f};
	const float dp[3] = {px.dpx0, px.dpx1, px.dpx2};
	float S = 0.f, Sd = 0.f;
#pragma unroll
	for (int ch = 0; ch < (DEPTH ? 4 : 3); ch++) {
		const float ar = st.last_alpha * st.last_color[ch] + (1.f - st.last_alpha) * st.acc_rec[ch];
		st.acc_rec[ch] = live ? ar : st.acc_rec[ch];
		st.last_color[ch] = live ? c4[ch] : st.last_color[ch];
		if (ch < 3) S += (c4[ch] - st.acc_rec[ch]) * dp[ch];
		else Sd = c4[ch] - st.acc_rec[ch];
	}
	st.last_alpha = live ? alpha : st.last_alpha;
	const float dL_dalpha = T * S + px.neg_Tfinal_bg * inv;
	const float ca = -2.0f * q0.z, cb = -q0.w, cc = -2.0f * q1.x;
	const float gdx = G * dx, gdy = G * dy;
	const float dG_ddelx = -gdx * ca - gdy * cb;
	const float dG_ddely = -gdy * cc - gdx * cb;
	float dL_dG = q1.z * dL_dalpha;
	v[0] = dL_dG * dG_ddelx * px.ddelx_dx;
	v[1] = dL_dG * dG_ddely * px.ddely_dy;
	v[2] = -0.5f * gdx * dx * dL_dG;
	v[3] = -0.5f * gdx * dy * dL_dG;
	v[4] = -0.5f * gdy * dy * dL_dG;
	v[5] = G * dL_dalpha;
	if (DEPTH) {   // the oracle's second pass: the same terms for dL_dalpha = T * (d_i - Rd)
		const float dLa = T * Sd;
		dL_dG = q1.z * dLa;
		v[0] += dL_dG * dG_ddelx * px.ddelx_dx;
		v[1] += dL_dG * dG_ddely * px.ddely_dy;
		v[2] += -0.5f * gdx * dx * dL_dG;
		v[3] += -0.5f * gdx * dy * dL_dG;
		v[4] += -0.5f * gdy * dy * dL_dG;
		v[5] += G * dLa;
	}
	const float aT = alpha * T;
	v[6] = aT * px.dpx0;
	v[7] = aT * px.dpx1;
	v[8] = aT * px.dpx2;
	v[9] = DEPTH ? aT * px.gz : 0.f;
}

// DEPTH = false: the reference's backward (dL_depths ignored).  DEPTH = true: the opt-in extension
// that also differentiates the normalised depth target (math in
// oracle/bsr_oracle.c:bsro_render_backward_depth): a tenth partial sum dL/dz per instance and one more
// term in dL/dalpha.  out_depth is the forward's depth image (its zeros are the acc <= 0.5 gate).
template <bool DEPTH>
__global__ void __launch_bounds__(BSR_BLOCK) k_render_bwd_strict(int n_tiles, int gx, int W, int H,
                                                                 const uint2* __restrict__ tile_range,
                                                                 const uint32_t* __restrict__ point_list,
                                                                 const float4* __restrict__ rec,
                                                                 const uint32_t* __restrict__ wg_base,
                                                                 const float* __restrict__ bg_color,
                                                                 const float* __restrict__ final_Ts,
                                                                 const uint32_t* __restrict__ n_contrib,
                                                                 const float* __restrict__ dL_dpixels,
                                                                 const float* __restrict__ out_depth,   // DEPTH only
                                                                 const float* __restrict__ dL_depths,   // DEPTH only
                                                                 const int* __restrict__ masks_flag,    // forward's hand-over word (flags[6]; flags[2] = kept instances)
                                                                 int capacity,                          // the R the call was handed
                                                                 float4* __restrict__ slab)        // [R][9 or 10 floats]
{
	constexpr int NV = DEPTH ? 10 : 9;
	__shared__ BwdShared<NV> sh;

	const int tile = xcd_tile(blockIdx.x, n_tiles);
	if (tile >= n_tiles) return;
	// more instances kept than the R this call was handed: an overflowed BSR_FLAG_NO_READBACK forward -- no lists exist
	// (k_preprocess_bwd writes NaN gradients, the thread's next forward reports it)
	if (__builtin_amdgcn_readfirstlane(masks_flag[-4]) > capacity) return;
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int tx = tile % gx, ty = tile / gx;
	const int px = tx * BSR_TILE + ((wave & 1) << 3) + (lane & 7);
	const int py = ty * BSR_TILE + ((wave >> 1) << 3) + (lane >> 3);
	const bool inside = px < W && py < H;
	const float pixfx = (float)px, pixfy = (float)py;
	const float tile_x0 = (float)(tx * BSR_TILE), tile_y0 = (float)(ty * BSR_TILE);
	const size_t pix_id = (size_t)W * py + px;
	const size_t plane = (size_t)H * W;

	const uint2 range = tile_range[tile];
	const uint32_t start = range.x;
	const int n = (int)(range.y - range.x);
	// (the forward may have left its box tests in the top byte of the point_list words: k_render_bwd_t; unused here)
	const uint32_t id_mask = __builtin_amdgcn_readfirstlane(*masks_flag) != 0 ? 0x00ffffffu : 0xffffffffu;

	const float T_final = inside ? final_Ts[pix_id] : 0.0f;
	PairState pst = {};
	pst.T = T_final;
	const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0u;
	PixelConst pc = {};
	if (inside) {
		pc.dpx0 = dL_dpixels[pix_id];
		pc.dpx1 = dL_dpixels[plane + pix_id];
		pc.dpx2 = dL_dpixels[2 * plane + pix_id];
	}
	const float bg_dot_dpixel = bg_color[0] * pc.dpx0 + bg_color[1] * pc.dpx1 + bg_color[2] * pc.dpx2;
	pc.neg_Tfinal_bg = -T_final * bg_dot_dpixel;
	// depth extension: d_i = gz * z_i + g1 plays the role of a fourth colour channel
	if (DEPTH && inside) {
		const float depth_px = out_depth[pix_id];
		if (depth_px != 0.0f) {   // the forward's acc > 0.5 decision
			pc.gz = dL_depths[pix_id] / (1e-6f + (1.0f - T_final));
			pc.g1 = -pc.gz * depth_px;
		}
	}
	// component whose wave total lands in this lane after the reduction; one lane per component stores
	bool stores;
	const int comp_of_lane = calibrate_components<DEPTH>(lane, stores);
	float* const part_mine = &sh.part[wave][stores ? comp_of_lane : 0][0];
	pc.ddelx_dx = (float)(0.5 * W);
	pc.ddely_dy = (float)(0.5 * H);

	// Entries at list positions >= max(last_contributor) are skipped by every pixel of the tile
	// (reference :498-500): start the walk at the deepest entry any pixel blended.
	uint32_t m = last_contributor;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
	if (lane == 0) sh.max_contrib[wave] = m;
	if (tid < BSR_BWD_BATCH) {
#pragma unroll
		for (int w = 0; w < 4; w++)
#pragma unroll
			for (int k = 0; k < NV; k++) sh.part[w][k][tid] = 0.f;
	}
	__syncthreads();
	const int n_walk = (int)max(max(sh.max_contrib[0], sh.max_contrib[1]), max(sh.max_contrib[2], sh.max_contrib[3]));

	for (int base = 0; base < n_walk; base += BSR_BWD_BATCH) {
		const int cnt = min(BSR_BWD_BATCH, n_walk - base);
		const int top = n_walk - 1 - base;   // list position of batch entry j is top - j
		const bool valid = tid < cnt;
		uint32_t my_row = 0;
		float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
		if (valid) {
			const uint32_t my_slot = start + (uint32_t)(top - tid);
			const uint32_t id = point_list[my_slot] & id_mask;
			const float4* r = rec + (size_t)id * BSR_REC;   // one 64-B line: record + rect + instance offset
			r0 = r[0];
			r1 = r[1];
			r2 = r[2];
			my_row = instance_index(wg_base, id, r2, r[3], tx, ty);   // the entry's row in the Gaussian-major slab
		}
		// (the trailing barrier of the previous iteration fenced the staging buffers)
		const int n_mine = stage_and_compact(sh.st, tid, valid, r0, r1, r2, tile_x0, tile_y0);
		const int n_u = __builtin_amdgcn_readfirstlane(n_mine);
		// entry j of the batch sits at list position top - j; this pixel blended positions < last_contributor
		// (reference :498-500): j > top - last_contributor, compared on the pre-scaled list offsets
		const int joff_min = (top - (int)last_contributor) * 16;
		auto visit = [&](const unsigned int joff) {
			const char* rec = stage_rec(sh.st, joff);
			const float4 q0 = rec_q0<BSR_BWD_BATCH>(rec);
			const float4 q1 = rec_q1<BSR_BWD_BATCH>(rec);   // conic c, power cut, opacity, depth
			const float dx = q0.x - pixfx;
			const float dy = q0.y - pixfy;
			const float power = (q0.z * dx * dx + q1.x * dy * dy) + q0.w * dx * dy;   // pre-scaled conic (common.h): the forward's bits
			const bool cand = ((int)joff > joff_min) && !(power > 0.0f) && !(power < q1.y);
			if (wave_ballot(cand) == 0ull) return;   // wave-uniform
			const float4 q2 = rec_q2<BSR_BWD_BATCH>(rec);
			// The forward decided `alpha >= 1/255` on alpha = min(0.99, o * E(power)) with the pinned exp E (bsr_expf) --
			// in its default mode only inside the decision band, outside of which every exp within ulps decides alike
			// (render_fwd.hip) -- and the backward must take the same decision on every pair (the T chain divides by the
			// same factors the forward multiplied).  Lanes that must not blend carry G = 0, hence alpha = 0: every
			// recurrence then leaves their state unchanged (T / 1 = T) and all nine contributions are exactly 0.
			float G = cand ? bsr_expf_walk(power) : 0.f;
			float alpha = fminf(0.99f, q1.z * G);
			const bool active = !(alpha < 1.0f / 255.0f);
			if (wave_ballot(active) == 0ull) return;
			G = active ? G : 0.f;
			alpha = active ? alpha : 0.f;
			float v[10];
			pair_terms_reference<DEPTH>(pst, pc, q0, q1, q2, dx, dy, G, alpha, v);
			const float tot = wave_sums_masked<DEPTH>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]);
			if (stores) *reinterpret_cast<float*>(reinterpret_cast<char*>(part_mine) + (joff >> 2)) = tot;   // part_mine[j]
		};
		// four list entries per trip: one address computation and one 16-byte LDS read for the list
		for (int i = 0; i < n_u; i += 4) {
			const uint4 l = *reinterpret_cast<const uint4*>(&sh.st.list[wave][i]);   // (reads past the end stay inside the list)
			visit(l.x);
			if (i + 1 < n_u) visit(l.y);
			if (i + 2 < n_u) visit(l.z);
			if (i + 3 < n_u) visit(l.w);
		}
		__syncthreads();
		if (valid) {
			float a9[10];
			a9[9] = 0.f;
#pragma unroll
			for (int k = 0; k < NV; k++) {   // fixed order over the 4 quadrants -> deterministic
				a9[k] = ((sh.part[0][k][tid] + sh.part[1][k][tid]) + sh.part[2][k][tid]) + sh.part[3][k][tid];
				sh.part[0][k][tid] = 0.f;
				sh.part[1][k][tid] = 0.f;
				sh.part[2][k][tid] = 0.f;
				sh.part[3][k][tid] = 0.f;
			}
			float* const row = reinterpret_cast<float*>(slab) + (size_t)my_row * slab_row_floats(DEPTH);
			*reinterpret_cast<bsr_f32x4_a4*>(row) = bsr_f32x4{a9[0], a9[1], a9[2], a9[3]};
			*reinterpret_cast<bsr_f32x4_a4*>(row + 4) = bsr_f32x4{a9[4], a9[5], a9[6], a9[7]};
			if (DEPTH) *reinterpret_cast<bsr_f32x2_a4*>(row + 8) = bsr_f32x2{a9[8], a9[9]};
			else row[8] = a9[8];
		}
		__syncthreads();
	}

	// entries no pixel of the tile reached: zero rows, but they still need their map entry
	for (int pos = n_walk + tid; pos < n; pos += BSR_BLOCK) {
		const uint32_t slot = start + (uint32_t)pos;
		const uint32_t id = point_list[slot] & id_mask;
		float* const row = reinterpret_cast<float*>(slab) +
		                   (size_t)instance_index(wg_base, id, rec[(size_t)id * BSR_REC + 2], rec[(size_t)id * BSR_REC + 3],
		                                          tx, ty) * slab_row_floats(DEPTH);
		const bsr_f32x4 z = {0.f, 0.f, 0.f, 0.f};
		*reinterpret_cast<bsr_f32x4_a4*>(row) = z;
		*reinterpret_cast<bsr_f32x4_a4*>(row + 4) = z;
		if (DEPTH) *reinterpret_cast<bsr_f32x2_a4*>(row + 8) = bsr_f32x2{0.f, 0.f};
		else row[8] = 0.f;
	}
}

// =====================================================================================================================
// k_render_bwd_t: the default backward walk.  Per-pixel decisions as the reference; the cross-lane reduction of the
// entry's nine (ten) sums is TRANSPOSED through LDS and runs at full lane utilisation:
//
//   staging  (64-entry batches; lane j <-> entry j): the waves take their lists -- one per 8 x 4 HALF of the wave's
//            quadrant with NS = 2 -- from the half masks the forward left in the entries' point_list words (top byte;
//            render_fwd.hip), by two ballots and prefix popcounts; only wave 0 gathers the records (one 64-B line each),
//            writes them to LDS and keeps the entries' slab rows.  One barrier.  (No masks -- more than 2^24 Gaussians,
//            a single-list forward, the "no_half_masks" hook --: every wave gathers the records and tests them itself.)
//   phase 1  (per visit; lane = pixel of the wave's 8x8 quadrant; the wave's two halves walk their own lists side by
//            side, so a visit works on two different entries and the wave needs max(upper, lower) visits instead of
//            their union): power, candidate votes, exp, alpha, the T chain, Srec, dL_dalpha -- and then only the TWO
//            numbers every one of the entry's sums is linear in,
//                gd = G dL_dalpha      (mean2D / conic / opacity sums = sum over pixels of gd x (1, dx, dy, dx dx, dx dy, dy dy))
//                aT = alpha T          (colour sums = sum over pixels of aT x dL_dpixel[ch])
//            written as one 8-byte LDS store to the visit's SLOT of the wave's private buffer.  No cross-lane
//            instruction, no per-pair products.
//   phase 2  (once per CHUNK of 8 visits; lane = (slot e = lane >> 3, pixel row r = lane & 7)): the lane reads its row's
//            8 x (gd, aT) with four 16-byte LDS reads and forms the row's moments of gd about ITS entry's centre (the
//            reference's d = centre - pixel: dy one number per row, dx_c = dx_0 - c; moments about the tile centre
//            shifted afterwards cancel 50-100x on small splats) and the three colour sums against dL_dpixel of its own 8
//            pixels (register constants).  NS = 2: a slot's rows 0-3 and 4-7 belong to two entries, so the rows are added
//            over the quad (18 DPP adds) and the quad's four lanes ADD their shares to part[wave][k][entry] -- upper
//            halves first, lower halves behind them, plain read-add-write (an entry in both lists gets two
//            contributions; 0 + a + b does not depend on their order).  NS = 1: 19 DPP adds over the 8-lane group, plain
//            stores.
//   epilogue (per batch): wave 0 adds the four quadrants' parts in a fixed order and writes the entry's 36-byte row of
//            the Gaussian-major slab, while the other waves already stage the next batch.
//
// Sums are formed in a fixed order (bit-reproducible); no atomics.  LDS 31.0 KB -> 5 workgroups per CU (the limit is
// 32 000 B each); amdgpu_waves_per_eu(5,5) holds the kernel at 96 VGPRs.  History, A/B tables and rejected variants
// (scalar-cache records, LDS atomics, sums filed by list position, pair-padded lists, blocks of four entries, shared
// staging, a trailing barrier, list words prefetched a batch ahead, ...): docs/EXPERIMENTS.md.
// The depth-gradient instantiation (ten sums per entry) keeps NS = 1: with split lists it spills.
#ifndef BSR_BWT_BATCH
#define BSR_BWT_BATCH 64
#endif
#define BSR_BWT_CHUNK 8
#define BSR_BWT_ILP 2   // list entries per straight-line block of phase 1 (blocks of four: +1 %, docs/EXPERIMENTS.md)
#define BSR_BWT_PAD 4   // the lists are sentinel-padded to whole TRIPS of four; a chunk is one or two trips (pairs / eights:
                        // docs/EXPERIMENTS.md)
// A slot = 8 pixel rows x 64 B.  Phase-1 store: ds_write_b64 is served in groups of 16 consecutive lanes = 2 rows = 32
// consecutive dwords: conflict-free.  Phase-2 read: ds_read_b128 is served in four groups of 16 lanes that mix slots
// (e, e+1, e+2, e+3) x 4 rows each; rows 16 dwords apart cover all 64 banks once per FOUR rows, so slot e is rotated by
// 4 e dwords (16 bytes): slot stride 528 B makes every group hit 64 distinct banks (MI355X_MICROARCH.md "LDS").
#define BSR_BWT_SLOT 528
#define BSR_BWT_ROW (BSR_BWT_BATCH + 1)
// NS = lists per quadrant: 1, or 2 = one per 8 x 4 half -- the wave's two halves then walk DIFFERENT entries side by side
// (see the kernel's header)
template <int NV, int NS>
struct BwtShared {
	TileStageS<BSR_BWT_BATCH, NS, unsigned int, BSR_BWT_PAD> st;   // per-quadrant (per-half) lists, sentinel-padded to whole blocks
	float part[4][NV][BSR_BWT_ROW];   // per-wave partial sums of the current batch (plain stores); column BATCH = the sentinel's
	uint32_t max_contrib[4];
	alignas(16) char tbuf[4][BSR_BWT_CHUNK * BSR_BWT_SLOT];   // per wave: [slot][row][col] x (gd, aT)
};

// Sum over the 8 lanes sharing (lane >> 3) of nine (ten) values.  On return, in every lane with bit 2 clear x0..x3 (x4
// with TEN) hold the totals of inputs 0..3 (0..4) and x8 that of input 8; in lanes with bit 2 set x0..x3 (x4) hold the
// totals of inputs 4..7 (5..9).  Step over lane bit 2 = halving on bank-masked row rotations (rotate by 12 = "from
// lane + 4" into banks 0 and 2, rotate by 4 = "from lane - 4" into banks 1 and 3: both stay inside the 8-lane group), steps
// over bits 1 and 0 = plain quad permutes.  19 (20) instructions.  Inline asm gets no automatic wait states: a DPP read
// needs 2 after a VALU write of the same VGPR -- the order keeps >= 2 instructions between, s_nop at the start.
template <bool TEN>
__device__ __forceinline__ void row8_sums(float& x0, float& x1, float& x2, float& x3, float& x4, float& x5, float& x6,
                                          float& x7, float& x8, float& x9)
{
	if (TEN) {
		asm volatile(
		    "s_nop 1\n"
		    "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %1, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %2, %2, %2 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %3, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %4, %4, %4 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %0, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %1, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %2, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %3, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %4, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9));
	} else {
		asm volatile(
		    "s_nop 1\n"
		    "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %1, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %2, %2, %2 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %3, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %8, %8, %8 row_ror:12 row_mask:0xf bank_mask:0x5\n"
		    "v_add_f32_dpp %0, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %1, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %2, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %3, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n"
		    "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9));
	}
}

// NS = 2: the 8 lanes of a slot hold TWO entries (rows 0-3 | rows 4-7): the sums stop at the quad.  On return every
// lane holds the totals of its quad (its half of the slot) in all nine (ten) registers.  18 (20) instructions.
template <bool TEN>
__device__ __forceinline__ void row4_sums(float& x0, float& x1, float& x2, float& x3, float& x4, float& x5, float& x6,
                                          float& x7, float& x8, float& x9)
{
#define BSR_QSTEP(P)                                                                   \
	"v_add_f32_dpp %0, %0, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n"           \
	"v_add_f32_dpp %1, %1, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n"           \
	"v_add_f32_dpp %2, %2, %2 quad_perm:" P " row_mask:0xf bank_mask:0xf\n"           \
	"v_add_f32_dpp %3, %3, %3 quad_perm:" P " row_mask:0xf bank_mask:0xf\n"           \
	"v_add_f32_dpp %4, %4, %4 quad_perm:" P " row_mask:0xf bank_mask:0xf\n"           \
	"v_add_f32_dpp %5, %5, %5 quad_perm:" P " row_mask:0xf bank_mask:0xf\n"           \
	"v_add_f32_dpp %6, %6, %6 quad_perm:" P " row_mask:0xf bank_mask:0xf\n"           \
	"v_add_f32_dpp %7, %7, %7 quad_perm:" P " row_mask:0xf bank_mask:0xf\n"           \
	"v_add_f32_dpp %8, %8, %8 quad_perm:" P " row_mask:0xf bank_mask:0xf\n"
	if (TEN) {
		asm volatile("s_nop 1\n" BSR_QSTEP("[2,3,0,1]") "v_add_f32_dpp %9, %9, %9 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
		             BSR_QSTEP("[1,0,3,2]") "v_add_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9));
	} else {
		asm volatile("s_nop 1\n" BSR_QSTEP("[2,3,0,1]") BSR_QSTEP("[1,0,3,2]")
		             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9));
	}
#undef BSR_QSTEP
}

#ifdef BSR_WALK_TIMELINE
// (make timeline: per-workgroup start / end stamps of the default backward walk; tools/walk_stats.py --timeline)
__device__ unsigned long long g_bwd_times[4 * 70000];
#endif
template <bool DEPTH, int NS>
// Occupancy target handed to the register allocator.  The kernel's LDS allows 5 workgroups per CU = 5 waves per SIMD = 96
// VGPRs; left alone the allocator takes 113 (4 waves).  A/B on one box: 5 waves -7.5 % (the walk waits on LDS round trips
// 40-55 % of a wave's cycles: occupancy is what hides them); 6 waves (80 VGPRs, 48-entry batches for the LDS) spills 17-29
// dwords into the inner loop: +25..+60 %.
#ifndef BSR_BWT_WAVES
#define BSR_BWT_WAVES 5
#endif
#define BSR_BWT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(BSR_BWT_WAVES, BSR_BWT_WAVES)))
__global__ void __launch_bounds__(BSR_BLOCK) BSR_BWT_WAVES_ATTR k_render_bwd_t(int n_tiles, int gx, int W, int H,
                                                            const uint2* __restrict__ tile_range,
                                                            const uint32_t* __restrict__ point_list,
                                                            const float4* __restrict__ rec,
                                                            const uint32_t* __restrict__ wg_base,
                                                            const float* __restrict__ bg_color,
                                                            const float* __restrict__ final_Ts,
                                                            const uint32_t* __restrict__ n_contrib,
                                                            const float* __restrict__ dL_dpixels,
                                                            const float* __restrict__ out_depth,   // DEPTH only
                                                            const float* __restrict__ dL_depths,   // DEPTH only
                                                            int* __restrict__ masks_flag,          // forward's hand-over word (flags[6]; flags[2] = kept instances)
                                                                 int capacity,                          // the R the call was handed
                                                            float4* __restrict__ slab)        // [R][9 or 10 floats]
{
	constexpr int NV = DEPTH ? 10 : 9;
	constexpr int B = BSR_BWT_BATCH;
	static_assert(B <= 64 && B % 4 == 0, "a wave stages a batch lane by lane");
	__shared__ BwtShared<NV, NS> sh;

	// more instances kept than the R this call was handed: an overflowed BSR_FLAG_NO_READBACK forward -- no lists exist
	// (k_preprocess_bwd writes NaN gradients, the thread's next forward reports it).  (Ahead of the pool draw: either
	// every workgroup of the launch draws or none does.)
	if (__builtin_amdgcn_readfirstlane(masks_flag[-4]) > capacity) return;
	__shared__ int s_slot;
	const int tile = pooled_tile(blockIdx.x, n_tiles, masks_flag + (BSR_POOL_BWD - 6), &s_slot);
	if (tile < 0) return;
#ifdef BSR_WALK_TIMELINE
	const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int tx = tile % gx, ty = tile / gx;
	const int px = tx * BSR_TILE + ((wave & 1) << 3) + (lane & 7);
	const int py = ty * BSR_TILE + ((wave >> 1) << 3) + (lane >> 3);
	const bool inside = px < W && py < H;
	const float pixfx = (float)px, pixfy = (float)py;
	const float tile_x0 = (float)(tx * BSR_TILE), tile_y0 = (float)(ty * BSR_TILE);
	const size_t pix_id = (size_t)W * py + px;
	const size_t plane = (size_t)H * W;

	const uint2 range = tile_range[tile];
	const uint32_t start = range.x;
	const int n = (int)(range.y - range.x);
	// The forward's split-list staging left its eight per-half box tests in the top byte of every point_list word it
	// staged (render_fwd.hip; a word the forward never staged lies beyond every pixel's last contributor): the waves then
	// take their lists from that byte and only wave 0 gathers records.  0: plain ids, every wave tests for itself.
	const bool has_masks = __builtin_amdgcn_readfirstlane(*masks_flag) != 0;
	const uint32_t id_mask = has_masks ? 0x00ffffffu : 0xffffffffu;

	const float T_final = inside ? final_Ts[pix_id] : 0.0f;
	float T = T_final, Srec = 0.f;
	const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0u;
	float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f;
	if (inside) {
		dpx0 = dL_dpixels[pix_id];
		dpx1 = dL_dpixels[plane + pix_id];
		dpx2 = dL_dpixels[2 * plane + pix_id];
	}
	const float bg_dot_dpixel = bg_color[0] * dpx0 + bg_color[1] * dpx1 + bg_color[2] * dpx2;
	const float neg_Tfinal_bg = -T_final * bg_dot_dpixel;
	float gz = 0.f, g1 = 0.f;   // depth extension: d_i = gz * z_i + g1 plays the role of a fourth colour channel
	if (DEPTH && inside) {
		const float depth_px = out_depth[pix_id];
		if (depth_px != 0.0f) {   // the forward's acc > 0.5 decision
			gz = dL_depths[pix_id] / (1e-6f + (1.0f - T_final));
			g1 = -gz * depth_px;
		}
	}
	const float ddelx_dx = (float)(0.5 * W);
	const float ddely_dy = (float)(0.5 * H);

	// ---- phase-2 identity of this lane: slot e2 = lane >> 3, pixel row r2 = lane & 7 of the wave's quadrant.  Its
	// constants: dL_dpixel (and gz) of the row's 8 pixels, handed over through the wave's (still idle) transposition
	// buffer -- lane p holds pixel p's; the row's y and the quadrant's first column as floats.
	char* const tb_w = sh.tbuf[wave];
	float cw0[8], cw1[8], cw2[8], cwz[DEPTH ? 8 : 1];
	{
		*reinterpret_cast<float4*>(tb_w + lane * 16) = make_float4(dpx0, dpx1, dpx2, gz);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const float4* rowp = reinterpret_cast<const float4*>(tb_w + (lane & 7) * 128);
#pragma unroll
		for (int c = 0; c < 8; c++) {
			const float4 v = rowp[c];
			cw0[c] = v.x;
			cw1[c] = v.y;
			cw2[c] = v.z;
			if (DEPTH) cwz[c] = v.w;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();   // (the buffer is reused by the walk)
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	const float py_row = (float)(ty * BSR_TILE + ((wave >> 1) << 3) + (lane & 7));   // phase 2: y of row r2
	const float px_q0 = (float)(tx * BSR_TILE + ((wave & 1) << 3));                    //          x of the quadrant's column 0
	char* const tb_store = tb_w + (lane >> 3) * 64 + (lane & 7) * 8;                  // phase 1: this pixel's place in slot 0
	const char* const tb_load = tb_w + (lane >> 3) * BSR_BWT_SLOT + (lane & 7) * 64;  // phase 2: row r2 of slot e2
	const bool lane_stores = (lane & 3) == 0;                            // lanes 0 and 4 of every 8-lane group
	// part[wave][k0 + m][j]: k0 = 0 for the lanes holding the totals of values 0..3 (0..4), 4 (5) for the others
	float* const part_k0 = &sh.part[wave][(lane & 4) ? (DEPTH ? 5 : 4) : 0][0];
	float* const part_q = &sh.part[wave][(lane & 3) == 0 ? 0 : (lane & 3) == 1 ? 3 : (lane & 3) == 2 ? 5 : 7][0];   // NS = 2
	const char* const rec0 = reinterpret_cast<const char*>(&sh.st.q0[0]);
	stage_init(sh.st, tid);   // the sentinel record (read behind the staging's barriers)

	// Entries at list positions >= max(last_contributor) are skipped by every pixel of the tile
	// (reference :498-500): start the walk at the deepest entry any pixel blended.
	uint32_t m = last_contributor;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
	if (lane == 0) sh.max_contrib[wave] = m;
	if (tid < BSR_BWT_ROW) {
#pragma unroll
		for (int w = 0; w < 4; w++)
#pragma unroll
			for (int k = 0; k < NV; k++) sh.part[w][k][tid] = 0.f;
	}
	__syncthreads();
	const int n_walk = (int)max(max(sh.max_contrib[0], sh.max_contrib[1]), max(sh.max_contrib[2], sh.max_contrib[3]));

	for (int base = 0; base < n_walk; base += B) {
		const int cnt = min(B, n_walk - base);
		const int top = n_walk - 1 - base;   // list position of batch entry j is top - j
		// ---- staging: EVERY wave looks at the batch's 64 entries itself (lane j <-> entry j: the same 64-B lines four
		// times, three of them from the L1/L2) and tests them against ITS OWN quadrant: one box test per wave instead of
		// four on one wave, its list straight from its own ballot -- no count table, one barrier instead of two.  Wave 0
		// also writes the records to LDS and keeps the entries' slab rows for the epilogue.
		// (the trailing barrier of the previous iteration fenced the staging buffers)
		const bool valid = lane < cnt;      // (every wave; `valid` of the epilogue below: wave 0's)
		uint32_t my_row = 0;
		bool hit = false;
		bool hit_lo = false;   // NS = 2: rows 4-7 of the quadrant (`hit`: rows 0-3)
		if (valid) {
			const uint32_t word = point_list[start + (uint32_t)(top - lane)];
			const uint32_t id = word & id_mask;
			if (wave == 0 || !has_masks) {
				const float4* r = rec + (size_t)id * BSR_REC;   // one 64-B line: record + rect + instance offset
				const float4 r0 = r[0], r1 = r[1];
				if (wave == 0) {
					float4 r2 = r[2];
					my_row = instance_index(wg_base, id, r2, r[3], tx, ty);   // the entry's row in the Gaussian-major slab
					r2.w = r1.y + 2.2e-3f;   // staged q2.w: just above the decision band, -ln(255 o) + 1.2e-3
					sh.st.q0[lane] = r0;
					sh.st.q1[lane] = r1;
					sh.st.q2[lane] = r2;
				}
				if (!has_masks) {
					const float ca = -2.0f * r0.z, cb = -r0.w, cc = -2.0f * r1.x;   // the record holds (-a/2, -b, -c/2)
					const bool pd = (ca > 0.0f) && (cc > 0.0f) && (ca * cc - cb * cb > 0.0f);
					const float qx = tile_x0 + (float)((wave & 1) << 3), qy = tile_y0 + (float)((wave >> 1) << 3);
					if (NS == 2) {
						hit = box_may_hit<7, 3>(r0.x, r0.y, ca, cb, cc, r1.y, -cb / cc, -cb / ca, pd, qx, qy);
						hit_lo = box_may_hit<7, 3>(r0.x, r0.y, ca, cb, cc, r1.y, -cb / cc, -cb / ca, pd, qx, qy + 4.0f);
					} else {
						hit = box_may_hit<7, 7>(r0.x, r0.y, ca, cb, cc, r1.y, -cb / cc, -cb / ca, pd, qx, qy);
					}
				}
			}
			if (has_masks) {   // the forward's tests of this entry against the two halves of this wave's quadrant
				const uint32_t two = (word >> (24 + 2 * wave)) & 3u;
				if (NS == 2) {
					hit = (two & 1u) != 0u;
					hit_lo = (two & 2u) != 0u;
				} else {
					hit = two != 0u;
				}
			}
		}
		int n_u;   // visits the wave needs for this batch (wave-uniform)
		if (NS == 2) {
			// one list per 8 x 4 half: the wave's two halves walk their own lists side by side (a visit then works on two
			// different entries), both padded with the sentinel to the longer one's length rounded up to whole blocks
			const unsigned long long hits = wave_ballot(hit), hits_lo = wave_ballot(hit_lo);
			const int n_hi = __builtin_amdgcn_readfirstlane((int)__popcll(hits));
			const int n_lo = __builtin_amdgcn_readfirstlane((int)__popcll(hits_lo));
			n_u = max(n_hi, n_lo);
			const int n_end = (n_u + (BSR_BWT_PAD - 1)) & ~(BSR_BWT_PAD - 1);
			const unsigned long long below = (1ull << lane) - 1ull;
			if (hit) sh.st.list[NS * wave][__popcll(hits & below)] = (unsigned int)(lane << 4);
			if (hit_lo) sh.st.list[NS * wave + (NS - 1)][__popcll(hits_lo & below)] = (unsigned int)(lane << 4);
			if (lane < n_end - n_hi) sh.st.list[NS * wave][n_hi + lane] = (unsigned int)(B << 4);
			if (lane < n_end - n_lo) sh.st.list[NS * wave + (NS - 1)][n_lo + lane] = (unsigned int)(B << 4);
		} else {
			const unsigned long long hits = wave_ballot(hit);
			n_u = __builtin_amdgcn_readfirstlane((int)__popcll(hits));
			if (hit) sh.st.list[wave][__popcll(hits & ((1ull << lane) - 1ull))] = (unsigned int)(lane << 4);
			if (lane < BSR_BWT_PAD - 1) sh.st.list[wave][n_u + lane] = (unsigned int)(B << 4);   // pad to whole blocks
		}
		__syncthreads();
		// entry j of the batch sits at list position top - j; this pixel blended positions < last_contributor
		// (reference :498-500): j > top - last_contributor, compared on the pre-scaled list offsets
		const int joff_min = (top - (int)last_contributor) * 16;

		// ---- phase 1: FOUR list entries per trip as one straight-line block (the forward walk's shape): the four
		// entries' LDS reads and their power / exp / alpha chains are independent and the scheduler interleaves them; only
		// T and Srec are serial.  No vote per entry, no early return: a lane that must not blend carries G = 0, hence
		// alpha = 0, which leaves its state unchanged (T / 1 = T, Srec + 0 S = Srec) and stores gd = aT = 0; the rows are
		// sentinel-padded to whole chunks and a sentinel is never a candidate.  Decisions exactly as k_render_bwd: the
		// forward decided `alpha >= 1/255` with the pinned exp; only a trip holding a candidate lane inside the decision
		// band (q2.w = its upper edge) evaluates the pinned exp and tests alpha, everywhere else the value comes from
		// v_exp_f32.  (Testing alpha on a trip's other entries changes nothing: outside the band every candidate has
		// alpha >= (1 + 1e-3) / 255.)
		struct Ent { float4 q0, q1, q2; float power; bool cand; uint64_t cand_mask; };
		auto load = [&](const unsigned int joff) {
			Ent e;
			const char* rec = rec0 + joff;
			e.q0 = srec_q0<B>(rec);      // x, y, -a/2, -b
			e.q1 = srec_q1<B>(rec);      // -c/2, power cut, opacity, depth
			e.q2 = srec_q2<B>(rec);      // r, g, b, upper edge of the decision band
			if (!DEPTH) asm volatile("" : : "v"(e.q1.w));   // keep the read one ds_read_b128 (4 LDS cycles; narrowed to b96 it costs 8)
			const float dx = e.q0.x - pixfx;
			const float dy = e.q0.y - pixfy;
			e.power = (e.q0.z * dx * dx + e.q1.x * dy * dy) + e.q0.w * dx * dy;   // pre-scaled conic (common.h): the forward's bits
			e.cand = ((int)joff > joff_min) && !(e.power > 0.0f) && !(e.power < e.q1.y);
			// the same predicate as a lane mask, assembled on the scalar side from ballots of SINGLE compares (a ballot of
			// an AND of lane masks makes hipcc materialise the mask in a VGPR and compare it again: two vector slots)
			e.cand_mask = wave_ballot((int)joff > joff_min) & wave_ballot(!(e.power > 0.0f)) & wave_ballot(!(e.power < e.q1.y));
			return e;
		};
		auto chain = [&](const Ent& e, float G, float alpha, const int slot) {
			float gd, aT;
			{
#pragma clang fp contract(fast)
				const float om = 1.f - alpha;
				const float inv = __builtin_amdgcn_rcpf(om);   // 1 ulp
				const float qT = T * inv;
				T = __builtin_fmaf(__builtin_fmaf(-om, qT, T), inv, qT);   // T / (1 - alpha), residual-corrected (see k_render_bwd)
				float u = e.q2.x * dpx0 + e.q2.y * dpx1 + e.q2.z * dpx2;
				if (DEPTH) u += __builtin_fmaf(gz, e.q1.w, g1);
				const float S = u - Srec;
				Srec = Srec + alpha * S;
				const float dL_dalpha = T * S + neg_Tfinal_bg * inv;
				gd = G * dL_dalpha;
				aT = alpha * T;
			}
			*reinterpret_cast<float2*>(tb_store + slot * BSR_BWT_SLOT) = make_float2(gd, aT);
		};
		// NE entries as one straight-line block
		auto block = [&](const unsigned int (&l)[BSR_BWT_ILP], const int slot0) {
			constexpr int NE = BSR_BWT_ILP;
			Ent e[NE];
			float g[NE], a[NE];
			uint64_t vote = 0ull;
#pragma unroll
			for (int k = 0; k < NE; k++) {
				e[k] = load(l[k]);
				g[k] = __builtin_amdgcn_exp2f(e[k].power * 1.44269504088896341f);
				// a candidate below the band's upper edge (or a NaN edge / power: opacity <= 0 or NaN) -> decide with the pinned exp
				vote |= e[k].cand_mask & wave_ballot(!(e[k].power >= e[k].q2.w));
			}
			const bool decide = vote != 0ull;
			if (decide) {   // rare (a few % of the trips): each lane picks its exp with the FORWARD's expression
#pragma unroll
				for (int k = 0; k < NE; k++)
					g[k] = !(fabsf(e[k].power - (e[k].q1.y + 1.0e-3f)) >= 1.1e-3f) ? bsr_expf_walk(e[k].power) : g[k];
			}
#pragma unroll
			for (int k = 0; k < NE; k++) {
				g[k] = e[k].cand ? g[k] : 0.f;
				a[k] = fminf(0.99f, e[k].q1.z * g[k]);
			}
			if (decide) {
#pragma unroll
				for (int k = 0; k < NE; k++) {
					const bool keep = !(a[k] < 1.0f / 255.0f);
					g[k] = keep ? g[k] : 0.f;
					a[k] = keep ? a[k] : 0.f;
				}
			}
#pragma unroll
			for (int k = 0; k < NE; k++) chain(e[k], g[k], a[k], slot0 + k);
		};
		auto trip = [&](const uint4 l, const int slot0) {
			const unsigned int q0[2] = {l.x, l.y}, q1[2] = {l.z, l.w};
			block(q0, slot0);
			block(q1, slot0 + 2);
		};

		// ---- phase 2: the chunk's slots, transposed: lane (e2, r2) sums row r2 of slot e2 about the ENTRY's centre
		auto reduce_chunk = [&](const int i0, const int n_valid) {
			// the slot's entry (possibly the sentinel).  Beyond n_valid the list holds whatever earlier batches left there (or
			// nothing yet): clamped, so that the reads below stay inside the staged records; those lanes store nothing
			// (NS = 2: rows 4-7 of the slot belong to the other list's entry)
			const unsigned int joff = min(sh.st.list[NS * wave + (NS == 2 ? ((lane >> 2) & 1) : 0)][i0 + (lane >> 3)], (unsigned int)(B << 4));
			const float2 xy = *reinterpret_cast<const float2*>(rec0 + joff);
			const float4 A = *reinterpret_cast<const float4*>(tb_load);        // g0 a0 g1 a1
			const float4 Bv = *reinterpret_cast<const float4*>(tb_load + 16);  // g2 a2 g3 a3
			const float4 C = *reinterpret_cast<const float4*>(tb_load + 32);   // g4 a4 g5 a5
			const float4 D = *reinterpret_cast<const float4*>(tb_load + 48);   // g6 a6 g7 a7
			float v0, v1, v2, v3, v4, v5, v6, v7, v8, v9 = 0.f;
			{
#pragma clang fp contract(fast)
				// d = centre - pixel, as the reference forms it (:501): the row's dy is one number, dx_c = dx_0 - c
				const float dy = xy.y - py_row;
				const float x0 = xy.x - px_q0, x1 = x0 - 1.0f, x2 = x0 - 2.0f, x3 = x0 - 3.0f, x4 = x0 - 4.0f, x5 = x0 - 5.0f,
				            x6 = x0 - 6.0f, x7 = x0 - 7.0f;
				const float t0 = A.x * x0, t1 = A.z * x1, t2 = Bv.x * x2, t3 = Bv.z * x3, t4 = C.x * x4, t5 = C.z * x5,
				            t6 = D.x * x6, t7 = D.z * x7;                                       // g_c dx_c
				const float s0 = ((A.x + A.z) + (Bv.x + Bv.z)) + ((C.x + C.z) + (D.x + D.z));   // sum g
				const float s1 = ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + (t6 + t7));             // sum g dx
				const float s2 = ((t0 * x0 + t1 * x1) + (t2 * x2 + t3 * x3)) + ((t4 * x4 + t5 * x5) + (t6 * x6 + t7 * x7));   // sum g dx dx
				v0 = s1;            // sum gd dx
				v1 = dy * s0;       // sum gd dy
				v2 = s2;            // sum gd dx dx
				v3 = dy * s1;       // sum gd dx dy
				v4 = dy * v1;       // sum gd dy dy
				v5 = s0;            // sum gd
				v6 = A.y * cw0[0] + A.w * cw0[1] + Bv.y * cw0[2] + Bv.w * cw0[3] + C.y * cw0[4] + C.w * cw0[5] + D.y * cw0[6] + D.w * cw0[7];
				v7 = A.y * cw1[0] + A.w * cw1[1] + Bv.y * cw1[2] + Bv.w * cw1[3] + C.y * cw1[4] + C.w * cw1[5] + D.y * cw1[6] + D.w * cw1[7];
				v8 = A.y * cw2[0] + A.w * cw2[1] + Bv.y * cw2[2] + Bv.w * cw2[3] + C.y * cw2[4] + C.w * cw2[5] + D.y * cw2[6] + D.w * cw2[7];
				if (DEPTH)
					v9 = A.y * cwz[0] + A.w * cwz[1] + Bv.y * cwz[2] + Bv.w * cwz[3] + C.y * cwz[4] + C.w * cwz[5] + D.y * cwz[6] + D.w * cwz[7];
			}
			if (NS == 2) {
				// the slot's two halves belong to two entries: sums over the quad only, and the quad's four lanes ADD their
				// shares of the nine (ten) totals to the entry's partial sums.  An entry that is in both of the wave's lists
				// gets two contributions, from two slots that may sit in one chunk: the upper halves' lanes read, add and
				// write first, the lower halves' behind them (a wave's LDS operations are served in order), so no two lanes
				// of one instruction share an address.  A partial sum is 0 + a (+ b), and a + b = b + a exactly: the order of
				// the two contributions does not show.  (LDS float atomics instead: the kernel takes 0.85 ms instead of 0.49.)
				row4_sums<DEPTH>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9);
				const int q = lane & 3;   // lane q of the quad adds components {0,1,2}, {3,4(,9)}, {5,6}, {7,8}
				const float wa = q == 0 ? v0 : q == 1 ? v3 : q == 2 ? v5 : v7;
				const float wb = q == 0 ? v1 : q == 1 ? v4 : q == 2 ? v6 : v8;
				float* const p = reinterpret_cast<float*>(reinterpret_cast<char*>(part_q) + (joff >> 2));   // part[wave][k_q][j]
				const bool live = (lane >> 3) < n_valid;   // (a sentinel slot's zeros land in the rows' pad column BATCH)
#pragma unroll
				for (int hh = 0; hh < 2; hh++) {
					if (live && ((lane >> 2) & 1) == hh) {
						p[0] += wa;
						p[BSR_BWT_ROW] += wb;
						if (q == 0) p[2 * BSR_BWT_ROW] += v2;
						if (DEPTH && q == 1) p[6 * BSR_BWT_ROW] += v9;
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				}
				return;
			}
			row8_sums<DEPTH>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9);
			if (lane_stores && (lane >> 3) < n_valid) {   // (a sentinel slot's zeros land in the rows' pad column BATCH)
				float* const p = reinterpret_cast<float*>(reinterpret_cast<char*>(part_k0) + (joff >> 2));   // part[wave][k0][j]
				p[0] = v0;
				p[BSR_BWT_ROW] = v1;
				p[2 * BSR_BWT_ROW] = v2;
				p[3 * BSR_BWT_ROW] = v3;
				if (DEPTH) p[4 * BSR_BWT_ROW] = v4;
				else if (!(lane & 4)) p[8 * BSR_BWT_ROW] = v8;
			}
		};

#ifdef BSR_BWT_KO_WALK   // knock-out build (timing only, results wrong): staging + epilogue without the walk
		const int n_pad = 0;
#else
		const int n_pad = (n_u + (BSR_BWT_PAD - 1)) & ~(BSR_BWT_PAD - 1);   // (the row is sentinel-padded to whole trips)
#endif
		const unsigned int* const my_list = &sh.st.list[NS * wave + (NS == 2 ? (lane >> 5) : 0)][0];   // NS = 2: rows 0-3 | rows 4-7 of the quadrant
		for (int i = 0; i < n_pad; i += BSR_BWT_CHUNK) {
			const uint4 la = *reinterpret_cast<const uint4*>(my_list + i);
			const uint4 lb = *reinterpret_cast<const uint4*>(my_list + i + 4);   // (past the padding: inside the struct, not used)
			trip(la, 0);
			if (i + 4 < n_pad) trip(lb, 4);
#ifdef BSR_BWT_KO_PHASE2   // knock-out build (timing only): phase 1 without the transposed reduction
			continue;
#endif
			// the wave's own stores, then its own loads: the LDS serves a wave's operations in order
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			reduce_chunk(i, min(BSR_BWT_CHUNK, n_pad - i));
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
		__syncthreads();
		if (valid && tid < B) {   // (wave 0: it holds the entries' slab rows)
			float a9[10];
			a9[9] = 0.f;
#pragma unroll
			for (int k = 0; k < NV; k++) {   // fixed order over the 4 quadrants -> deterministic
				a9[k] = ((sh.part[0][k][tid] + sh.part[1][k][tid]) + sh.part[2][k][tid]) + sh.part[3][k][tid];
				sh.part[0][k][tid] = 0.f;
				sh.part[1][k][tid] = 0.f;
				sh.part[2][k][tid] = 0.f;
				sh.part[3][k][tid] = 0.f;
			}
			// the wave sums -> the reference's sums (see k_render_bwd); this thread staged entry `tid` itself
			const float4 e0 = sh.st.q0[tid], e1 = sh.st.q1[tid];   // (x, y, -a/2, -b), (-c/2, cut, o, depth)
			const float ca = -2.0f * e0.z, cb = -e0.w, cc = -2.0f * e1.x;
			const float no = -e1.z;
			const float h = 0.5f * no;
			float* const row = reinterpret_cast<float*>(slab) + (size_t)my_row * slab_row_floats(DEPTH);
			*reinterpret_cast<bsr_f32x4_a4*>(row) = bsr_f32x4{no * ddelx_dx * (ca * a9[0] + cb * a9[1]),
			                                                   no * ddely_dy * (cc * a9[1] + cb * a9[0]), h * a9[2], h * a9[3]};
			*reinterpret_cast<bsr_f32x4_a4*>(row + 4) = bsr_f32x4{h * a9[4], a9[5], a9[6], a9[7]};
			if (DEPTH) *reinterpret_cast<bsr_f32x2_a4*>(row + 8) = bsr_f32x2{a9[8], a9[9]};
			else row[8] = a9[8];
		}
		// No barrier here: while wave 0 adds up the batch, waves 1-3 already stage the next one.  Safe ONLY while all of
		// these hold:  (1) only wave 0 writes st.q0 / q1 / q2 and reads or zeroes part[][];  (2) a wave's lists and its
		// tbuf are written and read by that wave alone;  (3) no wave writes part[] before the next staging barrier, which
		// wave 0 reaches behind this epilogue.  An epilogue shared between the waves, or any wave but 0 touching the staged
		// records ahead of the barrier, breaks them.  `make debug_variants` puts the barrier back
		// (libbsr_trailing_barrier.so); tests/test_round5_gpu.py compares the two builds bit for bit.
#ifdef BSR_BWT_TRAILING_BARRIER
		__syncthreads();
#endif
	}

	// entries no pixel of the tile reached: zero rows, but they still need their map entry
	for (int pos = n_walk + tid; pos < n; pos += BSR_BLOCK) {
		const uint32_t slot = start + (uint32_t)pos;
		const uint32_t id = point_list[slot] & id_mask;
		float* const row = reinterpret_cast<float*>(slab) +
		                   (size_t)instance_index(wg_base, id, rec[(size_t)id * BSR_REC + 2], rec[(size_t)id * BSR_REC + 3],
		                                          tx, ty) * slab_row_floats(DEPTH);
		const bsr_f32x4 z = {0.f, 0.f, 0.f, 0.f};
		*reinterpret_cast<bsr_f32x4_a4*>(row) = z;
		*reinterpret_cast<bsr_f32x4_a4*>(row + 4) = z;
		if (DEPTH) *reinterpret_cast<bsr_f32x2_a4*>(row + 8) = bsr_f32x2{0.f, 0.f};
		else row[8] = 0.f;
	}
#ifdef BSR_WALK_TIMELINE
	if (tid == 0 && tile < 70000) {   // (wave 0 leaves last or nearly so: it owns every batch's epilogue; filed under the tile)
		unsigned long long* t = g_bwd_times + 4 * (size_t)tile;
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

void launch_render_bwd(int gx, int gy, int W, int H, const uint2* tile_range, const uint32_t* point_list,
                       const float4* rec, const uint32_t* wg_base, const float* bg, const float* final_T,
                       const uint32_t* n_contrib, const float* dL_dpix, const float* out_depth, const float* dL_depths,
                       int* masks_flag, float4* slab, bool strict, int num_rendered, hipStream_t s)
{
	const int n_tiles = gx * gy;
	const int blocks = ((n_tiles + 7) / 8) * 8;
	const unsigned pad = occupancy_sweep_lds_pad("BSR_SWEEP_LDS_PAD_BWD");
	const bool depth = out_depth && dL_depths;
#define BSR_LAUNCH_STRICT(D_)                                                                                            \
	hipLaunchKernelGGL((k_render_bwd_strict<D_>), dim3(blocks), dim3(BSR_BLOCK), pad, s, n_tiles, gx, W, H, tile_range,     \
	                   point_list, rec, wg_base, bg, final_T, n_contrib, dL_dpix, depth ? out_depth : nullptr,           \
	                   depth ? dL_depths : nullptr, masks_flag, num_rendered, slab)
#define BSR_LAUNCH_BWT(D_, N_)                                                                                             \
	hipLaunchKernelGGL((k_render_bwd_t<D_, N_>), dim3(pooled_grid(n_tiles)), dim3(BSR_BLOCK), pad, s, n_tiles, gx, W, H,      \
	                   tile_range, point_list, rec, wg_base, bg, final_T, n_contrib, dL_dpix, depth ? out_depth : nullptr,   \
	                   depth ? dL_depths : nullptr, masks_flag, num_rendered, slab)
	// Default: the transposed-reduction walk, for every frame.  (Until round 5 frames with > 1900 reference instances
	// per tile -- C5, scales x 3 -- kept round 3's per-visit network walk, 0-4 % faster there; with the forward's half
	// masks handed over, k_render_bwd_t is 5 % faster at C5 and within 1.5 % on the dense scene: docs/EXPERIMENTS.md.)
	// BSR_FLAG_EXACT_GRAD: k_render_bwd_strict.
	if (strict) {
		if (depth) BSR_LAUNCH_STRICT(true);
		else BSR_LAUNCH_STRICT(false);
	} else if (depth) {
#ifndef BSR_BWT_DEPTH_NS
#define BSR_BWT_DEPTH_NS 1
#endif
		BSR_LAUNCH_BWT(true, BSR_BWT_DEPTH_NS);    // (ten sums per entry: with split lists the kernel spills and loses 8 %)
	} else {
		BSR_LAUNCH_BWT(false, 2);
	}
#undef BSR_LAUNCH_STRICT
#undef BSR_LAUNCH_BWT
}

}  // namespace bsr

#ifdef BSR_WALK_TIMELINE
extern "C" int bsr_debug_walk_timeline_bwd(void* out, size_t bytes)
{
	hipError_t e = hipDeviceSynchronize();
	if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(bsr::g_bwd_times), bytes);
	return e == hipSuccess ? 0 : 1;
}
#endif
