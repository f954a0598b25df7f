// Test-only: the product's PURE device functions, compiled from the product's own headers and run on given inputs, so
// that the python restatements of tests/test_box_test_cpu.py, test_tile_pool_cpu.py and test_bin_elem_cpu.py are checked
// against the code the kernels execute (an edit to box_may_hit, pooled_tile or the element packing would
// otherwise leave those CPU tests green).  Built by tests/native/Makefile into libbsr_pure_functions.so; driven by
// tests/test_round6_gpu.py through ctypes.  Nothing here is part of the product library.
#include "../../bloomscene_amd/csrc/tile_common.h"

using namespace bsr;

// out[i] = box_may_hit<EXT, EXTY>(...) with the derived arguments formed exactly as the callers form them
// (tile_common.h: stage_and_compact_s; preprocess.hip: the exact tile cull)
template <int EXT, int EXTY>
__global__ void k_box(int n, const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ a,
                      const float* __restrict__ b, const float* __restrict__ c, const float* __restrict__ cut,
                      const float* __restrict__ bx, const float* __restrict__ by, unsigned char* __restrict__ out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const bool pd = (a[i] > 0.0f) && (c[i] > 0.0f) && (a[i] * c[i] - b[i] * b[i] > 0.0f);
	const float rb_c = -b[i] / c[i], rb_a = -b[i] / a[i];
	out[i] = box_may_hit<EXT, EXTY>(X[i], Y[i], a[i], b[i], c[i], cut[i], rb_c, rb_a, pd, bx[i], by[i]) ? 1 : 0;
}

// element packing: store_elem_m -> load_elem_m / elem_tile_m / elem_key_m through a scratch buffer of 12 B per element
__global__ void k_elem(int n, int compact, const uint32_t* __restrict__ tile, const uint32_t* __restrict__ id,
                       const uint32_t* __restrict__ depth, BinElem* __restrict__ scratch, uint32_t* __restrict__ o_tile,
                       uint32_t* __restrict__ o_id, uint32_t* __restrict__ o_depth, uint32_t* __restrict__ o_tile_only,
                       unsigned long long* __restrict__ o_key)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	store_elem_m(scratch, (size_t)i, BinElem{tile[i], id[i], depth[i]}, compact);
	__threadfence();
	const BinElem e = load_elem_m(scratch, (size_t)i, compact);
	o_tile[i] = e.x; o_id[i] = e.y; o_depth[i] = e.z;
	o_tile_only[i] = elem_tile_m(scratch, (size_t)i, compact);
	o_key[i] = elem_key_m(scratch, (size_t)i, compact);
}

// the tile walks' workgroup -> tile assignment: every workgroup of a pooled launch records what pooled_tile hands it
__global__ void __launch_bounds__(256) k_pool(int n_tiles, int* pool_ctr, int* __restrict__ tile_of_wg)
{
	__shared__ int s_slot;
	const int t = pooled_tile((int)blockIdx.x, n_tiles, pool_ctr, &s_slot);
	if (threadIdx.x == 0) tile_of_wg[blockIdx.x] = t;
}

extern "C" {

int pt_box_may_hit(int ext, int exty, int n, const float* X, const float* Y, const float* a, const float* b, const float* c,
                   const float* cut, const float* bx, const float* by, unsigned char* out, void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	const dim3 g((n + 255) / 256), blk(256);
	if (ext == 15 && exty == 15) hipLaunchKernelGGL((k_box<15, 15>), g, blk, 0, s, n, X, Y, a, b, c, cut, bx, by, out);
	else if (ext == 7 && exty == 7) hipLaunchKernelGGL((k_box<7, 7>), g, blk, 0, s, n, X, Y, a, b, c, cut, bx, by, out);
	else if (ext == 7 && exty == 3) hipLaunchKernelGGL((k_box<7, 3>), g, blk, 0, s, n, X, Y, a, b, c, cut, bx, by, out);
	else return 1;
	return hipGetLastError() == hipSuccess ? 0 : 2;
}

int pt_elem_roundtrip(int n, int compact, const uint32_t* tile, const uint32_t* id, const uint32_t* depth, void* scratch,
                      uint32_t* o_tile, uint32_t* o_id, uint32_t* o_depth, uint32_t* o_tile_only, unsigned long long* o_key,
                      void* stream)
{
	hipLaunchKernelGGL(k_elem, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, compact, tile, id, depth,
	                   (BinElem*)scratch, o_tile, o_id, o_depth, o_tile_only, o_key);
	return hipGetLastError() == hipSuccess ? 0 : 2;
}

// (host side: no GPU) the bucket of every depth of a segment, as rank_sort computes it
void pt_rank_sort_buckets(int n, const float* z, float zmin, float zmax, int nb, unsigned int* out)
{
	const float scale = rank_sort_scale(zmin, zmax, nb);
	for (int i = 0; i < n; i++) out[i] = rank_sort_bucket(z[i], zmin, scale, nb);
}
int pt_pooled_grid(int n_tiles) { return pooled_grid(n_tiles); }
int pt_pool_tiles_per_band(int n_tiles) { return pool_tiles_per_band(n_tiles); }

// tile_of_wg: pt_pooled_grid(n_tiles) ints; pool_ctr: one zeroed int
int pt_pooled_tiles(int n_tiles, int* pool_ctr, int* tile_of_wg, void* stream)
{
	hipLaunchKernelGGL(k_pool, dim3(pooled_grid(n_tiles)), dim3(256), 0, (hipStream_t)stream, n_tiles, pool_ctr, tile_of_wg);
	return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
