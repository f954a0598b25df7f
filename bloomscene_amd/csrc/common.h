// Shared declarations for the gfx950 rasterizer kernels.  Built with -ffp-contract=off: every
// fp32 expression is evaluated in source order; the only fused multiply-adds are the explicit
// fmaf() calls in bsr_expf, so forward results can be compared bit for bit with the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BSR_TILE 16          // reference cuda_rasterizer/config.h:15-16
#define BSR_BLOCK 256        // one 16x16 tile per workgroup = 4 wave64
#define BSR_NEAR 0.2f        // reference cuda_rasterizer/auxiliary.h:154
#define BSR_PACK_MAX_SRC 8    // tensors bsr_pack_rows lays side by side

namespace bsr {

// Private layout of the three scratch buffers (opaque to callers).
// Per-Gaussian "splat record": everything the tile kernels gather per list entry, exactly one
// 64-byte cache line:
//   q0 = (x, y, -conic.a / 2, -conic.b)   q1 = (-conic.c / 2, power_cut, opacity, depth)   q2 = (r, g, b, 0)
//   q3 = bits(inst_offset (relative to its workgroup's base), xmin | ymin << 16, width | height << 16, kept_mask_lo), q2.w = bits(kept_mask_hi)
//        (backward only).  kept_mask bit k = the k-th tile (row-major) of the rect is kept; rects of
//        more than 64 tiles keep every tile and ignore the mask.
#define BSR_REC 4
struct GeomState {
	float4* rec;        // [P][BSR_REC]
	uint32_t* inst_offset;  // [P] start of the Gaussian's block of kept tile instances, relative to wg_base[id / 256]
	uint32_t* wg_kept;      // [ceil(P/256)] kept instances per preprocess workgroup; exclusive-scanned in place
	                        //               into the workgroup's base by k_scan_tiles (-> "wg_base")
	uint32_t* wg_area;      // [ceil(P/256)] rect tiles per preprocess workgroup (sum = reference num_rendered)
	uint32_t* hist1;        // [256][8 * ceil(n_wg / 8)] + [256] + [256]: kept instances per (low 8 bits of the tile id, preprocess
	                        //   workgroup), digit-major, row-scanned in place by k_scans; then the digit totals (k_scans) and
	                        //   their exclusive prefix, the digit bases (k_emit_scatter).
	                        //   First radix pass = k_emit_scatter: instances are written straight to their pass-1 place.
	uint64_t* kept_mask;    // [P] see q3 above
	ushort4* rect;      // [P] tile rect (xmin, ymin, xmax, ymax); zero area <=> culled
	uint8_t* clamped;   // [P] bit ch = SH colour channel ch was clamped at 0
	float* depth;           // [P] view-space depth again, compact: k_emit_scatter needs nothing else of the 64-B record
	static size_t bytes(size_t P);
	static GeomState carve(char* p, size_t P);
};
// One tile instance in the radix passes: 12 bytes, 4-byte aligned, moved as one dwordx3 access
// (load_elem / store_elem: a plain struct copy is split into dwordx2 + dword).
struct BinElem {
	uint32_t x;   // tile id
	uint32_t y;   // Gaussian id
	uint32_t z;   // bits of the view-space depth
};
// Slab rows (k_render_bwd -> k_preprocess_bwd) are TIGHT: 9 floats per kept instance (10 with the depth-gradient
// extension), 4-byte aligned, moved with wide accesses that only assume that alignment.
typedef float bsr_f32x4 __attribute__((ext_vector_type(4)));
typedef float bsr_f32x2 __attribute__((ext_vector_type(2)));
typedef bsr_f32x4 bsr_f32x4_a4 __attribute__((aligned(4)));
typedef bsr_f32x2 bsr_f32x2_a4 __attribute__((aligned(4)));
__host__ __device__ __forceinline__ constexpr int slab_row_floats(bool depth_grad) { return depth_grad ? 10 : 9; }
#define BSR_SLAB_ROW_BYTES 40   // what the binning buffer reserves per instance (the wider row) ...
#define BSR_SLAB_TAIL_BYTES 16  // ... plus the 12 B a reader's last 16-byte load may reach past a run
typedef uint32_t bsr_u32x3 __attribute__((ext_vector_type(3)));
typedef bsr_u32x3 bsr_u32x3_a4 __attribute__((aligned(4)));
__device__ __forceinline__ BinElem load_elem(const BinElem* p)
{
	const bsr_u32x3 v = *reinterpret_cast<const bsr_u32x3_a4*>(p);
	return BinElem{v.x, v.y, v.z};
}
__device__ __forceinline__ void store_elem(BinElem* p, const BinElem e)
{
	*reinterpret_cast<bsr_u32x3_a4*>(p) = bsr_u32x3{e.x, e.y, e.z};
}
// Compact form of an element in memory (8 bytes instead of 12) for the common case -- tile ids of up to 16 bits (the
// tile-owned second binning pass) and Gaussian ids below 2^24: word 0 = tile id bits 8..15 << 24 | Gaussian id, word 1 =
// depth bits.  The low tile byte is the pass-1 bucket the element lies in, which every reader knows.  In registers an
// element is a BinElem either way (x = the tile id, its low byte zero when read back from the compact form).  `compact`
// is uniform over the launch.  The buffers are sized for the 12-byte form.
__device__ __forceinline__ BinElem load_elem_m(const BinElem* base, size_t i, int compact)
{
	if (compact) {
		const uint2 v = reinterpret_cast<const uint2*>(base)[i];
		return BinElem{(v.x >> 24) << 8, v.x & 0x00ffffffu, v.y};
	}
	return load_elem(base + i);
}
__device__ __forceinline__ void store_elem_m(BinElem* base, size_t i, const BinElem e, int compact)
{
	if (compact) reinterpret_cast<uint2*>(base)[i] = make_uint2(((e.x >> 8) << 24) | e.y, e.z);
	else store_elem(base + i, e);
}
// the element's tile id alone (low byte zero in the compact form)
__device__ __forceinline__ uint32_t elem_tile_m(const BinElem* base, size_t i, int compact)
{
	return compact ? (reinterpret_cast<const uint2*>(base)[i].x >> 24) << 8 : base[i].x;
}
// its sort key inside a tile: (depth bits, Gaussian id)
__device__ __forceinline__ uint64_t elem_key_m(const BinElem* base, size_t i, int compact)
{
	if (compact) {
		const uint2 v = reinterpret_cast<const uint2*>(base)[i];
		return ((uint64_t)v.y << 32) | (uint64_t)(v.x & 0x00ffffffu);
	}
	const BinElem e = load_elem(base + i);
	return ((uint64_t)e.z << 32) | (uint64_t)e.y;
}
#define BSR_HIST_BLOCKS_MAX 2048
struct BinState {
	uint32_t* point_list; // [R] gaussian ids, tile-major, (depth, id)-sorted  (first: the backward needs only this); once
	                      //     k_render_fwd has staged an entry (one view, ids < 2^24, split lists): id | half mask << 24
	BinElem* elems_a;     // [R] (tile id, gaussian id, depth bits): ping-pong buffers of the radix passes
	BinElem* elems_b;       // [R]
	float4* slab;         // [R][9 or 10 floats, tight] the backward's per-instance partial sums (k_render_bwd -> k_preprocess_bwd): the SAME
	                      //        bytes as elems_a / elems_b, which are dead once the forward has returned
	uint32_t* hist;       // [256 * BSR_HIST_BLOCKS_MAX] digit-major workgroup histograms, then [256] digit totals
	static size_t bytes(size_t R, bool with_slab);
	static BinState carve(char* p, size_t R, bool with_slab);
};
struct ImgState {
	float* final_T;        // [N]
	uint32_t* n_contrib;   // [N]
	uint2* tile_range;     // [T] ranges[t] = [x, y) in point_list (the reference's `ranges`, rasterizer_impl.h:47).  The segments
	                       //     tile the first `kept` entries of point_list, in tile order or -- after k_bucket_sort -- in
	                       //     (low tile byte, high tile byte) order: nobody reads across a segment's ends
	int* flags;            // [BSR_FLAGS_BYTES / 4]: prefiltered violation | #tiles (1024, 4096] | kept instances | rect tiles (= reference
	                       //      num_rendered) | #tiles (4096, 8192] | #tiles > 8192 | point_list words carry the forward's
	                       //      per-half box tests in their top byte (k_render_fwd -> k_render_bwd*) | - | ... |
	                       //      [64], [96]: pool counters of k_render_fwd / k_render_bwd_t (pooled_tile below)
	uint32_t* big_tiles;   // [3][T] tiles with more than 1024 instances, one list per size class (any order):
	                       //        work lists of the wide sort kernels
	static size_t bytes(size_t N, size_t T);
	static ImgState carve(char* p, size_t N, size_t T);
};

// Kernel argument blocks (passed by value).
struct PreArgs {
	int P, D, M;
	const float* means3D;
	const float* scales;
	float scale_modifier;
	const float* rotations;
	const float* opacities;
	const float* shs;
	const float* cov3D_precomp;
	const float* colors_precomp;
	const float* viewmatrix;
	const float* projmatrix;
	const float* cam_pos;
	int W, H;
	float tan_fovx, tan_fovy, focal_x, focal_y;
	int gx, gy;
	int prefiltered;
	int* radii;          // may be NULL; [n_views][P]
	int n_views;         // >= 1.  View-batched forward (bsr_forward_views): grid.y = view, viewmatrix / projmatrix /
	                     // cam_pos hold n_views entries, every geometry row is indexed by the virtual id
	                     // view * P_pad + idx (P_pad = P rounded up to 256) and tile rows are offset by view * gy:
	                     // the views are stacked into one virtual image of n_views * gy tile rows
	GeomState geom;
	int* flags;            // [0] prefiltered violation, [2] kept instances, [3] rect tiles (both set by k_scans)
};

struct BwdArgs {
	int P, D, M;
	const float* means3D;
	const int* radii;
	const float* shs;
	const float* scales;
	const float* rotations;
	float scale_modifier;
	const float* cov3D_precomp;
	const float* viewmatrix;
	const float* projmatrix;
	const float* campos;
	float tan_fovx, tan_fovy, focal_x, focal_y;
	GeomState geom;
	const float4* slab;            // [R][9 floats; 10 with depth_grad] per-instance partial sums written by k_render_bwd, Gaussian-major:
	                               //        row wg_base[g/256] + inst_offset[g] + k = k-th kept tile of Gaussian g
	int depth_grad;                // extension: slab rows carry a tenth float, dL/d(view z), added to dL_dmean3D
	const int* kept_ptr;           // flags[2] of the forward: kept tile instances (device)
	int capacity;                  // the R this backward was handed.  kept > capacity <=> a BSR_FLAG_NO_READBACK forward
	                               // overflowed: nothing behind the counters is valid, the call writes NaN gradients
	float* dL_dmean2D;         // [P,3]  (outputs; fully written)
	float* dL_dconic;          // [P,4]
	float* dL_dopacity;        // [P]
	float* dL_dcolor;          // [P,3]
	float* dL_dmean3D;         // [P,3]
	float* dL_dcov3D;          // [P,6]
	float* dL_dsh;             // [P,M,3]
	float* dL_dscale;          // [P,3]
	float* dL_drot;            // [P,4]
};

// Records the calling thread's error message (read back by bsr_last_error) and returns 1.  api.hip
int fail(const char* fmt, ...);

// One blocking 4-byte device->host read through the calling thread's pinned landing buffer.  api.hip
int read_u32_blocking(const uint32_t* dev, uint32_t* out, hipStream_t s);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) & ~(a - 1); }

// ---- pinned exp: identical algorithm to bsro_expf in oracle/bsr_oracle.c (<= 1 ulp) ----
__device__ __forceinline__ float bsr_expf(float x)
{
	if (x != x) return x;
	if (x < -104.0f) return 0.0f;
	if (x > 88.72284f) return __builtin_inff();
	const float k = __builtin_rintf(x * 1.44269504088896341f);
	float r = __builtin_fmaf(-k, 0.693359375f, x);
	r = __builtin_fmaf(-k, -2.12194440e-4f, r);
	float p = 1.0f / 5040.0f;
	p = __builtin_fmaf(p, r, 1.0f / 720.0f);
	p = __builtin_fmaf(p, r, 1.0f / 120.0f);
	p = __builtin_fmaf(p, r, 1.0f / 24.0f);
	p = __builtin_fmaf(p, r, 1.0f / 6.0f);
	p = __builtin_fmaf(p, r, 0.5f);
	p = __builtin_fmaf(p, r, 1.0f);
	p = __builtin_fmaf(p, r, 1.0f);
	return __builtin_ldexpf(p, (int)k);
}

// Wave-wide vote on a predicate, straight from the compare's lane mask.  (HIP's __ballot(int)
// materialises the predicate as 0/1 in a VGPR and compares it again: two VALU ops per vote, which
// matters inside the tile walks.)
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Inclusive sum over the 64 lanes on the vector ALU alone (the sequence LLVM's atomic optimiser emits for wave64 on
// GFX9): four row_shr steps inside the 16-lane rows, then lane 15 of rows 0 / 2 broadcast into rows 1 / 3 and lane 31
// into rows 2 - 3.  (__shfl_up is a ds_bpermute round trip per step.)
__device__ __forceinline__ uint32_t wave_inclusive_sum_dpp(uint32_t x)
{
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
	return x;
}

// Branch-free bsr_expf for the tile walks, without the range checks (NaN flows through the fma chain):
// there the result is only consumed on lanes whose argument passed `power <= 0` and
// `power >= power_cut`, and power_cut = -ln(255 o) - 1e-3
// is above -104 for every finite opacity (-ln(255 * FLT_MAX) = -94.3), so on those lanes bsr_expf's
// underflow cut never fires; other lanes' results are discarded by the callers' predication.  Bit-identical to
// bsr_expf on every consumed lane.
__device__ __forceinline__ float bsr_expf_walk(float x)
{
	const float k = __builtin_rintf(x * 1.44269504088896341f);
	float r = __builtin_fmaf(-k, 0.693359375f, x);
	r = __builtin_fmaf(-k, -2.12194440e-4f, r);
	float p = 1.0f / 5040.0f;
	p = __builtin_fmaf(p, r, 1.0f / 720.0f);
	p = __builtin_fmaf(p, r, 1.0f / 120.0f);
	p = __builtin_fmaf(p, r, 1.0f / 24.0f);
	p = __builtin_fmaf(p, r, 1.0f / 6.0f);
	p = __builtin_fmaf(p, r, 0.5f);
	p = __builtin_fmaf(p, r, 1.0f);
	p = __builtin_fmaf(p, r, 1.0f);
	return __builtin_ldexpf(p, (int)k);
}

// Dense index of rect tile k (row-major) among the Gaussian's kept tiles.
__device__ __forceinline__ uint32_t kept_rank(uint32_t area, uint64_t mask, uint32_t k)
{
	return area > 64u ? k : (uint32_t)__popcll(mask & ((1ull << k) - 1ull));
}
__device__ __forceinline__ bool tile_kept(uint32_t area, uint64_t mask, uint32_t k)
{
	return area > 64u || ((mask >> k) & 1ull);
}
__device__ __forceinline__ uint32_t kept_count(uint32_t area, uint64_t mask)
{
	return area > 64u ? area : (uint32_t)__popcll(mask);
}

// The bucket map of the per-tile bucket-and-rank sort (binning.hip: rank_sort): the depth word read as the float it is,
//     b = min(int((z - z_min) * scale), nb - 1),   scale = (nb - 0.5) / (z_max - z_min)   (0 when all depths are equal or the quotient overflows).
// Monotone in z whatever the roundings (a - c, x * s with s >= 0 and float -> int are all non-decreasing), and for
// positive finite floats z is monotone in the depth WORD: buckets in ascending order hold ascending keys.  Linear in
// the depth itself: a shift of the raw bits would be logarithmic, 8x denser at the far end of a 1 .. 10 range than at
// the near end.  Segments holding other depth patterns are not offered to this sort.
// (Host-callable: tests/native/pure_functions.hip checks both against tests/test_rank_sort_cpu.py.)
__host__ __device__ __forceinline__ float rank_sort_scale(float zmin, float zmax, int nb)
{
	const float s = zmax > zmin ? ((float)nb - 0.5f) / (zmax - zmin) : 0.0f;
	return s < 3.0e38f ? s : 0.0f;   // (a range of a few denormals: the quotient overflows -- one bucket, i.e. declined or trivial)
}
__host__ __device__ __forceinline__ uint32_t rank_sort_bucket(float z, float zmin, float scale, int nb)
{
	const uint32_t b = (uint32_t)((z - zmin) * scale);
	return b < (uint32_t)nb ? b : (uint32_t)(nb - 1);
}

// Column of preprocess workgroup `wg` in the digit-major pass-1 histogram: workgroups that run on the
// same XCD (wg, wg + 8, wg + 16, ...) are neighbours, so the 4-byte entries they write into one 64-byte
// sector merge in that XCD's L2.  (Any fixed order works: it only decides in which order the workgroups'
// buckets are laid out.)  `per` = ceil(n_wg / 8); columns whose workgroup does not exist count as zero.
__device__ __forceinline__ int hist1_column(int wg, int per) { return (wg & 7) * per + (wg >> 3); }
__device__ __forceinline__ int hist1_wg_of_column(int c, int per) { return (c % per) * 8 + c / per; }

// reference forward.cu:118-152 (quaternion NOT normalised, :127)
__device__ __forceinline__ void cov3d_from_scale_rot(const float* scale, float mod, const float4 q, float* cov3D)
{
	const float s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
	const float r = q.x, x = q.y, y = q.z, z = q.w;
	// M[c][k] = s_k * R[c][k]
	const float m00 = s0 * (1.f - 2.f * (y * y + z * z)), m01 = s1 * (2.f * (x * y - r * z)), m02 = s2 * (2.f * (x * z + r * y));
	const float m10 = s0 * (2.f * (x * y + r * z)), m11 = s1 * (1.f - 2.f * (x * x + z * z)), m12 = s2 * (2.f * (y * z - r * x));
	const float m20 = s0 * (2.f * (x * z - r * y)), m21 = s1 * (2.f * (y * z + r * x)), m22 = s2 * (1.f - 2.f * (x * x + y * y));
	cov3D[0] = m00 * m00 + m01 * m01 + m02 * m02;
	cov3D[1] = m10 * m00 + m11 * m01 + m12 * m02;
	cov3D[2] = m20 * m00 + m21 * m01 + m22 * m02;
	cov3D[3] = m10 * m10 + m11 * m11 + m12 * m12;
	cov3D[4] = m20 * m10 + m21 * m11 + m22 * m12;
	cov3D[5] = m20 * m20 + m21 * m21 + m22 * m22;
}


// XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2), so give each XCD a
// contiguous band of tiles; neighbouring tiles gather many of the same splat records.
__device__ __forceinline__ int xcd_tile(int b, int n_tiles)
{
	const int per = (n_tiles + 7) >> 3;
	return (b & 7) * per + (b >> 3);
}

// Tail pool of the default tile walks (k_render_fwd, k_render_bwd_t).  The hardware deals workgroups to the eight XCDs in
// turn whatever their speed, and the XCDs of one chip differ by a few per cent (per box: tools/walk_stats.py
// --timeline): with xcd_tile alone the slowest XCD ends 3-5 % (forward) / 2-3 % (backward) behind the mean.  So the
// last BSR_POOL_K tiles of every band are not owned by anybody: the grid carries BSR_POOL_K + BSR_POOL_E workgroups per
// XCD behind the band's owned part, and each of them draws its tile from ONE counter -- draw j = tile own + j / 8 of
// band j % 8 -- or leaves when the 8 K pool tiles are gone.  An XCD that is through with its owned tiles early draws
// more than K, a late one fewer; every workgroup still renders at most one tile (a resident grid looping over tiles
// was measured and rejected: docs/EXPERIMENTS.md).  The workgroup that makes the launch's last draw zeroes the counter
// (every pool workgroup draws exactly once), so a second backward on the same forward state finds it as k_scans left it.
// Counters: flags[BSR_POOL_FWD], flags[BSR_POOL_BWD], each on a 128-byte line of its own (a line shared with flags[2] /
// flags[6], which every tile reads, made tiles 1.7x slower in the resident-grid experiment).
#ifndef BSR_POOL_K
#define BSR_POOL_K 64
#endif
#ifndef BSR_POOL_E
#define BSR_POOL_E 48
#endif
#define BSR_FLAGS_BYTES 512
#define BSR_POOL_FWD 64
#define BSR_POOL_BWD 96
// 0: launches of less than ~2.5 rounds of resident workgroups (under 4096 tiles) -- there the draws cost more than the
// XCDs differ (800 x 800, 2500 tiles: k_render_fwd 0.025 -> 0.030 ms with the pool)
__host__ __device__ __forceinline__ int pool_tiles_per_band(int n_tiles)
{
	const int per = (n_tiles + 7) >> 3;
	return per >= 8 * BSR_POOL_K ? BSR_POOL_K : 0;
}
// workgroups of a pooled launch
static inline int pooled_grid(int n_tiles)
{
	const int per = (n_tiles + 7) >> 3;
	return 8 * (per + (pool_tiles_per_band(n_tiles) ? BSR_POOL_E : 0));
}
// The calling workgroup's tile, -1 = none.  ALL threads of the workgroup call this (one barrier on the pool path);
// s_slot: one int of LDS.
__device__ __forceinline__ int pooled_tile(int b, int n_tiles, int* pool_ctr, int* s_slot)
{
	const int per = (n_tiles + 7) >> 3;
	const int k = pool_tiles_per_band(n_tiles), own = per - k;
	const int i = b >> 3;
	int tile;
	if (i < own) {
		tile = (b & 7) * per + i;
	} else {
		if (threadIdx.x == 0) {
			const int j = atomicAdd(pool_ctr, 1);
			if (j == 8 * (k + BSR_POOL_E) - 1) *pool_ctr = 0;   // the launch's last draw
			*s_slot = j;
		}
		__syncthreads();
		const int j = __builtin_amdgcn_readfirstlane(*s_slot);
		tile = j < 8 * k ? (j & 7) * per + own + (j >> 3) : n_tiles;
	}
	tile = __builtin_amdgcn_readfirstlane(tile);   // (wave-uniform on both paths: keep everything derived from it scalar)
	return tile < n_tiles ? tile : -1;
}

}  // namespace bsr
