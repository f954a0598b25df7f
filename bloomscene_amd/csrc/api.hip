// C ABI (include/bloomscene_rast.h) and host orchestration of the gfx950 rasterizer.
//
// Stage order of one forward call (cf. the reference's CudaRasterizer::Rasterizer::forward,
// cuda_rasterizer/rasterizer_impl.cu:198-339):
//   k_preprocess (project, cull, SH, exact tile cull, Gaussian-major instance numbering)
//   -> k_scans (workgroup bases, totals, pass-1 histogram rows) -> 16-byte D2H read (kept, num_rendered), overlapped with:
//   binning alloc (sized from the previous call) -> k_emit_scatter (= first radix pass) -> k_tile_count ->
//   k_tile_starts -> k_tile_scatter (tile ids of up to 16 bits; else the remaining radix passes -> k_tile_ranges)
//   -> k_sort_tiles_* (per-tile LDS sort) -> k_render_fwd.  No float atomics anywhere.
// Backward (rasterizer_impl.cu:403-504): k_render_bwd (per-instance partial sums to a Gaussian-major
// slab, no atomics) -> k_preprocess_bwd (adds each Gaussian's adjacent rows, then the chain).
#include "../../include/bloomscene_rast.h"
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <vector>

namespace bsr {

static constexpr size_t BSR_RADIX_BINS_ = 256;

// ---------------------------------------------------------------- scratch layouts
size_t GeomState::bytes(size_t P)
{
	size_t s = 0;
	s += align_up(P * BSR_REC * sizeof(float4), 256);
	s += align_up(P * sizeof(uint32_t), 256);
	s += 2 * align_up(((P + 255) / 256) * sizeof(uint32_t), 256);
	s += align_up((256 * (((P + 255) / 256 + 7) / 8 * 8) + 512) * sizeof(uint32_t), 256);   // rows + digit totals + digit bases
	s += align_up(P * sizeof(uint64_t), 256);
	s += align_up(P * sizeof(ushort4), 256);
	s += align_up(P * sizeof(uint8_t), 256);
	s += align_up(P * sizeof(float), 256);
	return s + 256;
}
GeomState GeomState::carve(char* p, size_t P)
{
	GeomState g;
	p = (char*)align_up((size_t)p, 256);
	g.rec = (float4*)p;      p += align_up(P * BSR_REC * sizeof(float4), 256);
	g.inst_offset = (uint32_t*)p; p += align_up(P * sizeof(uint32_t), 256);
	g.wg_kept = (uint32_t*)p;     p += align_up(((P + 255) / 256) * sizeof(uint32_t), 256);
	g.wg_area = (uint32_t*)p;     p += align_up(((P + 255) / 256) * sizeof(uint32_t), 256);
	g.hist1 = (uint32_t*)p;       p += align_up((256 * (((P + 255) / 256 + 7) / 8 * 8) + 512) * sizeof(uint32_t), 256);
	g.kept_mask = (uint64_t*)p;   p += align_up(P * sizeof(uint64_t), 256);
	g.rect = (ushort4*)p;    p += align_up(P * sizeof(ushort4), 256);
	g.clamped = (uint8_t*)p; p += align_up(P * sizeof(uint8_t), 256);
	g.depth = (float*)p;
	return g;
}
// The backward's slab (sized at 40 B per instance + 16: its rows are 36 B, 40 B with the depth gradient, and the
// reader's last 16-byte load of a run may reach 12 B past it) lives in the caller's binning buffer too, over the radix ping-pong
// buffers, which are dead once the forward has returned: the library owns no device memory, as in the reference,
// where every byte of scratch comes from the caller's resize callbacks (rasterize_points.cu:27-33).
// with_slab = false (view-batched forward: inference only) sizes the section for the ping-pong buffers alone.
static size_t bin_work_bytes(size_t R, bool with_slab)
{
	const size_t pingpong = 2 * align_up(R * sizeof(BinElem), 256);
	const size_t slab = with_slab ? align_up(R * BSR_SLAB_ROW_BYTES + BSR_SLAB_TAIL_BYTES, 256) : 0;
	return pingpong > slab ? pingpong : slab;
}
size_t BinState::bytes(size_t R, bool with_slab)
{
	return align_up(R * sizeof(uint32_t), 256) + bin_work_bytes(R, with_slab) +
	       align_up((size_t)BSR_RADIX_BINS_ * (BSR_HIST_BLOCKS_MAX + 1) * sizeof(uint32_t), 256) + 256;
}
BinState BinState::carve(char* p, size_t R, bool with_slab)
{
	BinState b;
	p = (char*)align_up((size_t)p, 256);
	b.point_list = (uint32_t*)p; p += align_up(R * sizeof(uint32_t), 256);
	b.elems_a = (BinElem*)p;
	b.elems_b = (BinElem*)(p + align_up(R * sizeof(BinElem), 256));
	b.slab = (float4*)p;           p += bin_work_bytes(R, with_slab);
	b.hist = (uint32_t*)p;
	return b;
}
size_t ImgState::bytes(size_t N, size_t T)
{
	return 2 * align_up(N * 4, 256) + align_up(T * sizeof(uint2), 256) + BSR_FLAGS_BYTES + align_up(3 * T * 4, 256) + 256;
}
ImgState ImgState::carve(char* p, size_t N, size_t T)
{
	ImgState i;
	p = (char*)align_up((size_t)p, 256);
	i.final_T = (float*)p;        p += align_up(N * 4, 256);
	i.n_contrib = (uint32_t*)p;   p += align_up(N * 4, 256);
	i.tile_range = (uint2*)p;     p += align_up(T * sizeof(uint2), 256);
	i.flags = (int*)p;            p += BSR_FLAGS_BYTES;
	i.big_tiles = (uint32_t*)p;
	return i;
}

// ---------------------------------------------------------------- host-side cache for the forward's one read-back
// Pinned landing buffer + event for the asynchronous copy of the counters, and the shape / num_rendered
// of the previous forward call on this thread (the size guess of the next one).  Nothing here carries
// results between calls.
struct SyncCache {
	int* pinned = nullptr;       // [4] = flags[0..3] of the forward; [4] = landing word of read_u32_blocking;
	                             // [5] = error flag a prefiltered bsr_visible_filter kernel writes straight into host memory
	int* pinned_dev = nullptr;   // the same buffer as the device addresses it
	hipEvent_t copied = nullptr;
	hipEvent_t deferred = nullptr;   // behind the counters' copy of a BSR_FLAG_NO_READBACK forward (waited for by the NEXT call)
	int device = -1;
	int last_P = -1, last_W = -1, last_H = -1, last_V = -1;
	uint32_t last_R = 0;
	uint32_t last_kept = 0;      // kept instances of that call (the hint that picks the next call's binning plan)
	bool pending = false;        // a BSR_FLAG_NO_READBACK forward's copy of the counters is in flight / unchecked
	size_t pending_capacity = 0;
	int pending_P = -1, pending_W = -1, pending_H = -1;   // shape of that forward, and (once its counters have been
	int nr_P = -1, nr_W = -1, nr_H = -1;                  // checked) the shape and kept instances of the last checked one:
	uint32_t nr_kept = 0;                                 // the plan hint of the next no-readback forward of the same shape
};
static SyncCache* sync_cache()
{
	static thread_local SyncCache c;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { fail("hipGetDevice failed"); return nullptr; }
	if (c.device != dev) {   // first use on this thread, or the thread moved to another GPU
		if (c.copied) (void)hipEventDestroy(c.copied);
		if (c.deferred) (void)hipEventDestroy(c.deferred);
		c.copied = c.deferred = nullptr;
		c.pending = false;
		if (!c.pinned && hipHostMalloc((void**)&c.pinned, 8 * sizeof(int), hipHostMallocMapped) != hipSuccess) {
			c.pinned = nullptr;
			fail("hipHostMalloc failed");
			return nullptr;
		}
		if (hipHostGetDevicePointer((void**)&c.pinned_dev, c.pinned, 0) != hipSuccess) {
			c.pinned_dev = nullptr;
			fail("hipHostGetDevicePointer failed");
			return nullptr;
		}
		if (hipEventCreateWithFlags(&c.copied, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&c.deferred, hipEventDisableTiming) != hipSuccess) {
			c.copied = c.deferred = nullptr;
			fail("hipEventCreate failed");
			return nullptr;
		}
		c.device = dev;
		c.last_P = -1;
	}
	return &c;
}

// One 4-byte device->host read the caller blocks on (selection / visibility counts, anchors.hip): through the calling
// thread's PINNED landing buffer -- an asynchronous copy into pageable memory is staged by the runtime and signals
// later -- and waited for on an event.  Returns 0 on success.
int read_u32_blocking(const uint32_t* dev, uint32_t* out, hipStream_t s)
{
	SyncCache* sc = sync_cache();
	if (!sc) return 1;
	if (hipMemcpyAsync(sc->pinned + 4, dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
	    hipEventRecord(sc->copied, s) != hipSuccess || hipEventSynchronize(sc->copied) != hipSuccess)
		return fail("reading a count back failed: %s", hipGetErrorString(hipGetLastError()));
	*out = (uint32_t)sc->pinned[4];
	return 0;
}

// Deferred overflow check of the calling thread's last BSR_FLAG_NO_READBACK forward (include/bloomscene_rast.h).
static int check_deferred(SyncCache* sc)
{
	if (!sc->pending) return 0;
	sc->pending = false;
	if (hipEventSynchronize(sc->deferred) != hipSuccess)
		return fail("waiting for the counters of the previous no-readback forward failed: %s", hipGetErrorString(hipGetLastError()));
	const uint32_t kept = (uint32_t)sc->pinned[2];
	sc->nr_P = sc->pending_P; sc->nr_W = sc->pending_W; sc->nr_H = sc->pending_H;
	sc->nr_kept = kept;
	if ((size_t)kept > sc->pending_capacity)
		return fail("the previous BSR_FLAG_NO_READBACK forward of this thread kept %u tile instances but was given a capacity "
		            "of %zu: that frame was not rendered (NaN outputs); its num_rendered was %u",
		            kept, sc->pending_capacity, (uint32_t)sc->pinned[3]);
	return 0;
}

// ---------------------------------------------------------------- kernels (other translation units)
void launch_preprocess(const PreArgs& a, bool filter_only, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present, hipStream_t s);
void launch_visible_filter_views(int P, int V, const float* means3D, const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp, const float* viewmatrices,
                                 const float* projmatrices, int W, int H, float tan_fovx, float tan_fovy, int* radii,
                                 const int* group_of_view, int n_groups, uint8_t* group_mask, uint32_t* wg_counts,
                                 uint32_t* group_counts, hipStream_t s);
void launch_pack_rows(int R, int P, int n_src, const float* const* src, const int* widths, const int64_t* idx,
                      int idx_stride, float* dst_packed, float* const* dst_each, hipStream_t s);
void launch_scans(int n_wg, uint32_t* wg_kept, uint32_t* wg_area, int* flags, uint32_t* hist1, int* host_counts,
                  hipStream_t s);
int binning_plan(int P, int T, int capacity, long long kept_hint);
void launch_binning(int plan, int P, int T, int gx, const int* n_ptr, int capacity, const GeomState& geom, BinElem* elems_a,
                    BinElem* elems_b, uint32_t* hist, int hist_blocks_max, uint2* tile_range, uint32_t* big_tiles,
                    int* flags, BinElem** elems_sorted, BinElem** elems_free, int* compact_out, hipStream_t s);
void launch_sort_tiles(int plan, int T, int n_bound, const int* n_ptr, int capacity, uint2* tile_range,
                       const uint32_t* big_tiles, const int* flags, const uint32_t* digit_total1, const BinElem* elems,
                       BinElem* elems_free, uint32_t* point_list, int compact, int force_int, int small_grids, hipStream_t s);
void launch_render_fwd(int gx, int gy, int n_views, int W, int H, const int* n_ptr, int capacity, const uint2* tile_range,
                       uint32_t* point_list, int* masks_flag,
                       const float4* rec, const float* bg, float* final_T, uint32_t* n_contrib, float* out_color,
                       float* out_depth, bool exact_exp, bool nan_on_overflow, int* pool_ctr, hipStream_t s);
void launch_render_bwd(int gx, int gy, int W, int H, const uint2* tile_range, const uint32_t* point_list,
                       const float4* rec, const uint32_t* wg_base, const float* bg, const float* final_T,
                       const uint32_t* n_contrib, const float* dL_dpix, const float* out_depth, const float* dL_depths,
                       int* masks_flag, float4* slab, bool strict, int num_rendered, hipStream_t s);
void launch_preprocess_bwd(const BwdArgs& a, hipStream_t s);

// ---------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

int fail(const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return 1;
}

#define HIP_TRY(expr)                                                                              \
	do {                                                                                           \
		hipError_t _e = (expr);                                                                    \
		if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
	} while (0)

// After a launch: always catch launch errors; with debug also synchronise (reference CHECK_CUDA,
// cuda_rasterizer/auxiliary.h:166-173).
#define STAGE_CHECK(name, debug, stream)                                                           \
	do {                                                                                           \
		hipError_t _e = hipGetLastError();                                                         \
		if (_e == hipSuccess && (debug)) _e = hipStreamSynchronize(stream);                        \
		if (_e != hipSuccess) return fail("stage %s failed: %s", name, hipGetErrorString(_e));     \
	} while (0)

// ---------------------------------------------------------------- stage profiler (bench only)
struct StageRec {
	const char* name;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;   // pending (recorded, not yet read) pairs, oldest first
	size_t head = 0;
	double total_ms = 0;
	int launches = 0;
};
static bool g_prof_on = false;
static int g_prof_every = 1;                      // sample every Nth forward call (and the backward that follows it)
static char g_prof_only[32] = "";                 // non-empty: only this stage is bracketed (bsr_profile_only)
static std::atomic<unsigned> g_prof_calls{0};     // forward calls since enable
static std::atomic<bool> g_prof_this_call{true};  // whether the current forward/backward pair is sampled (process-wide:
                                                  // PyTorch runs the backward on an autograd worker thread)
static std::mutex g_prof_mu;
static std::vector<StageRec> g_stages;
static std::vector<hipEvent_t> g_free_events;   // recycled events (creating thousands of events is slow)

// Fold every pair whose end event has completed into the totals and recycle its events.
// wait == true blocks on unfinished ones (used by bsr_profile_read).
static void drain_stage(StageRec& r, bool wait)
{
	while (r.head < r.ev.size()) {
		auto& p = r.ev[r.head];
		if (wait) {
			if (hipEventSynchronize(p.second) != hipSuccess) break;
		} else if (hipEventQuery(p.second) != hipSuccess) {
			(void)hipGetLastError();   // hipErrorNotReady is not an error
			break;
		}
		float ms = 0;
		if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
			r.total_ms += ms;
			r.launches++;
		}
		g_free_events.push_back(p.first);
		g_free_events.push_back(p.second);
		r.head++;
	}
	if (r.head == r.ev.size()) {
		r.ev.clear();
		r.head = 0;
	}
}

static hipEvent_t take_event()
{
	if (!g_free_events.empty()) {
		hipEvent_t e = g_free_events.back();
		g_free_events.pop_back();
		return e;
	}
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess) return nullptr;
	return e;
}

struct StageTimer {
	hipStream_t s;
	StageRec* rec = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	StageTimer(const char* name, hipStream_t stream) : s(stream)
	{
		if (!g_prof_on) return;
		if (!strcmp(name, "preprocess")) g_prof_this_call = (g_prof_calls.fetch_add(1) % (unsigned)g_prof_every) == 0;
		if (!g_prof_this_call) return;
		if (g_prof_only[0] && strcmp(g_prof_only, name)) return;
		std::lock_guard<std::mutex> lk(g_prof_mu);
		for (auto& r : g_stages)
			if (r.name == name || !strcmp(r.name, name)) rec = &r;
		if (!rec) {
			if (g_stages.size() >= BSR_PROFILE_MAX_STAGES) return;
			g_stages.reserve(BSR_PROFILE_MAX_STAGES);
			g_stages.push_back(StageRec{name});
			rec = &g_stages.back();
		}
		// (pending pairs are folded in by bsr_profile_read / _reset, outside any timed region: reading eight pairs here
		// -- hipEventElapsedTime resolves timestamps on a slow path -- cost one step in 32 of a long bench run 3-4 ms;
		// only a run that never reads keeps the backlog bounded this way)
		if (rec->ev.size() - rec->head >= 4096) drain_stage(*rec, false);
		e0 = take_event();
		e1 = take_event();
		if (!e0 || !e1) { rec = nullptr; return; }
		(void)hipEventRecord(e0, s);
	}
	~StageTimer()
	{
		if (!rec) return;
		(void)hipEventRecord(e1, s);
		std::lock_guard<std::mutex> lk(g_prof_mu);
		rec->ev.emplace_back(e0, e1);
	}
};

}  // namespace bsr

using namespace bsr;

static int check_common(int P, int width, int height, const float* means3D, const float* scales, const float* rotations,
                        const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix);
// One or several views of the same Gaussians.  V = 1 is bsr_forward (scratch layouts as the backward expects them);
// V > 1 stacks the views into one virtual image of V * gy tile rows (see PreArgs::n_views) so that every kernel
// after k_preprocess runs unchanged over V * T tiles -- the sparse views of a camera sweep are launch/latency
// bound one by one.
static int forward_impl(int V, bsr_alloc_fn geometryBuffer, void* geometry_user, bsr_alloc_fn binningBuffer,
                        void* binning_user, bsr_alloc_fn imageBuffer, void* image_user, int P, int D, int M,
                        const float* background, int width,
                int height, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii,
                int debug, void* stream, int* num_rendered, unsigned flags)
{
	g_err[0] = 0;
	hipStream_t s = (hipStream_t)stream;
	// (BSR_FLAG_EXACT_GRAD concerns the backward alone: accepted here so that a caller can hand one word to both)
	const unsigned known_flags = BSR_FLAG_EXACT_EXP | BSR_FLAG_EXACT_GRAD | BSR_FLAG_NO_READBACK | BSR_FLAG_TEST_MASK;
	if (flags & ~known_flags) {
		if (num_rendered) *num_rendered = 0;
		return fail("forward: unknown flag bits 0x%x", flags & ~known_flags);
	}
	const bool no_readback = (flags & BSR_FLAG_NO_READBACK) != 0;
	const long long given_capacity = (no_readback && num_rendered) ? (long long)*num_rendered : 0;
	if (num_rendered) *num_rendered = 0;
	if (no_readback) {
		if (V != 1) return fail("BSR_FLAG_NO_READBACK is for single-view calls");
		if (prefiltered) return fail("BSR_FLAG_NO_READBACK cannot be combined with prefiltered (its violation flag is part of the read-back)");
		if (!num_rendered || given_capacity <= 0)
			return fail("BSR_FLAG_NO_READBACK: *num_rendered must hold the capacity (tile instances, > 0) on entry");
	}
	{   // the deferred status of this thread's previous no-readback forward comes first
		SyncCache* sc0 = sync_cache();
		if (!sc0) return 1;
		if (sc0->pending) {
			// the check is a blocking host wait on an event: inside a stream capture that would invalidate the capture
			// (or hang, depending on the capture mode) -- say so instead
			hipStreamCaptureStatus cs0 = hipStreamCaptureStatusNone;
			if (hipStreamIsCapturing(s, &cs0) != hipSuccess) { (void)hipGetLastError(); cs0 = hipStreamCaptureStatusNone; }
			if (cs0 != hipStreamCaptureStatusNone)
				return fail("forward during stream capture while the overflow check of this thread's previous BSR_FLAG_NO_READBACK "
				            "forward is pending: call bsr_check_deferred() before hipStreamBeginCapture");
		}
		if (check_deferred(sc0)) return 1;
	}
	if (no_readback) *num_rendered = (int)given_capacity;   // what the backward must be handed as R (same carve)
	if (P == 0) {   // reference rasterize_points.cu:68-82: zero images, no scratch, num_rendered = 0
		if (width <= 0 || height <= 0 || !out_color || !out_depth) return fail("invalid image outputs");
		HIP_TRY(hipMemsetAsync(out_color, 0, (size_t)V * 3 * width * height * sizeof(float), s));
		HIP_TRY(hipMemsetAsync(out_depth, 0, (size_t)V * width * height * sizeof(float), s));
		return 0;
	}
	if (check_common(P, width, height, means3D, scales, rotations, cov3D_precomp, viewmatrix, projmatrix)) return 1;
	if (!geometryBuffer || !binningBuffer || !imageBuffer) return fail("scratch allocation callback is null");
	if (!out_color || !out_depth || !background) return fail("out_color/out_depth/background is null");
	if (P > 0) {
		if (!opacities) return fail("opacities is null");
		if ((shs == nullptr) == (colors_precomp == nullptr))
			return fail("Please provide excatly one of either SHs or precomputed colors!");
		if (shs && (!cam_pos || M <= 0)) return fail("SH colours need cam_pos and M > 0");
		if (shs && D >= 0 && (D + 1) * (D + 1) > M && D <= 3)
			return fail("sh_degree %d needs %d coefficients per Gaussian, shs holds %d", D, (D + 1) * (D + 1), M);
	}
	const int gx = (width + BSR_TILE - 1) / BSR_TILE, gy = (height + BSR_TILE - 1) / BSR_TILE;
	if ((long long)V * gy > 65535 || (long long)V * gx * gy > 0x3fffffff) return fail("too many views for one call");
	const int T = gx * gy * V;   // tiles of the stacked virtual image
	const size_t N = (size_t)width * height * (size_t)V;
	const int n_wg = ((P + 255) / 256) * V;
	// geometry rows: P for one view (what the backward carves), V * P_pad virtual ids for several
	const size_t P_rows = V == 1 ? (size_t)P : (size_t)n_wg * 256;
	if (P_rows > 0x7fffffffu) return fail("too many (view, Gaussian) pairs for one call");

	char* geom_p = geometryBuffer(geometry_user, GeomState::bytes(P_rows));
	char* img_p = imageBuffer(image_user, ImgState::bytes(N, (size_t)T));
	if (!geom_p || !img_p) return fail("scratch allocation callback returned null");
	GeomState geom = GeomState::carve(geom_p, P_rows);
	ImgState img = ImgState::carve(img_p, N, (size_t)T);

	// flags[1..7] are initialised by k_scans (which precedes every kernel that counts into or reads them); flags[0] is
	// written by k_preprocess itself, and only for prefiltered calls, so only those pay a memset launch
	if (prefiltered) HIP_TRY(hipMemsetAsync(img.flags, 0, sizeof(int), s));

	{
		PreArgs a;
		memset(&a, 0, sizeof(a));
		a.P = P; a.D = D; a.M = M;
		a.means3D = means3D; a.scales = scales; a.scale_modifier = scale_modifier; a.rotations = rotations;
		a.opacities = opacities; a.shs = shs; a.cov3D_precomp = cov3D_precomp; a.colors_precomp = colors_precomp;
		a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.cam_pos = cam_pos;
		a.W = width; a.H = height; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
		a.focal_y = height / (2.0f * tan_fovy);   // reference rasterizer_impl.cu:223-224
		a.focal_x = width / (2.0f * tan_fovx);
		a.gx = gx; a.gy = gy; a.prefiltered = prefiltered; a.radii = radii; a.geom = geom;
		a.flags = img.flags;
		a.n_views = V;
		if (V > 1)   // sparse views: k_preprocess then writes only the non-zero bins of its pass-1 histogram
			HIP_TRY(hipMemsetAsync(geom.hist1, 0, (size_t)256 * ((size_t)(n_wg + 7) / 8 * 8) * sizeof(uint32_t), s));
		{
			StageTimer t("preprocess", s);
			launch_preprocess(a, false, s);
		}
		STAGE_CHECK("preprocess", debug, s);
	}
	SyncCache* sc = sync_cache();
	if (!sc) return 1;
	// k_scans writes the four counters straight into this thread's pinned, device-mapped landing buffer (no copy on the
	// stream) -- unless the stream is capturing: a graph must not carry a pointer into a host thread's buffer, and an
	// event recorded into a graph cannot be waited for on the host anyway
	bool capturing = false;
	if (no_readback) {
		hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
		if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
		capturing = cs != hipStreamCaptureStatusNone;
	}
	{
		StageTimer t("scan_wg", s);
		launch_scans(n_wg, geom.wg_kept, geom.wg_area, img.flags, geom.hist1, capturing ? nullptr : sc->pinned_dev, s);
	}
	STAGE_CHECK("scan_wg", debug, s);

	// flags[2] = instances kept after the exact tile cull, flags[3] = the reference's num_rendered
	// (sum of rect areas, rasterizer_impl.cu:278-282), which sizes the binning scratch -> host: the one
	// blocking read of the forward pass (the reference has the same one, rasterizer_impl.cu:282).
	//
	// The read is overlapped with the REST of the forward: binning, tile sort and render take the instance
	// count from device memory, so when the previous call on this thread had the same (P, width, height)
	// the scratch is sized from its num_rendered (+25 %, decaying slowly after a large view) BEFORE the
	// read, all remaining kernels are enqueued behind the copy, and the host only waits for the copy's
	// event (to return num_rendered).  If the guess was too small those kernels returned without touching
	// anything and the tail is simply run again with the exact size.
	const bool guess = sc->last_P == P && sc->last_W == width && sc->last_H == height && sc->last_V == V && sc->last_R > 0;
	size_t cap = 0;
	BinState bin;
	BinElem* elems_sorted = nullptr;
	BinElem* elems_free = nullptr;
	int elems_compact = 0;   // the binning wrote 8-byte elements (common.h: load_elem_m)
	// bins, sorts and renders with scratch sized for `capacity` instances; every kernel takes the real count
	// from device memory and returns at once if it exceeds the capacity
	auto run_tail = [&](size_t capacity, bool rerun, long long kept_hint) -> int {
		char* bin_p = binningBuffer(binning_user, BinState::bytes(capacity, V == 1));
		if (!bin_p) return fail("scratch allocation callback returned null");
		bin = BinState::carve(bin_p, capacity, V == 1);
		cap = capacity;
		if (rerun) {
			// A first tail that was NOT skipped (kept <= its capacity, but the backward's carve would not fit) has already
			// filed its long tiles: flags[1], [4], [5] count the work lists of the wide sort classes and are zeroed only by
			// k_scans.  Counting again on top of them would list every long tile twice (two workgroups sorting one tile
			// through the same global scratch) and could spill one class's list into the next.
			HIP_TRY(hipMemsetAsync(img.flags + 1, 0, sizeof(int), s));
			HIP_TRY(hipMemsetAsync(img.flags + 4, 0, 2 * sizeof(int), s));
		}
		const int* n_ptr = img.flags + 2;
		const int plan = binning_plan((int)P_rows, T, (int)capacity, kept_hint);
		{
			StageTimer t("binning", s);
			launch_binning(plan, (int)P_rows, T, gx, n_ptr, (int)capacity, geom, bin.elems_a, bin.elems_b, bin.hist, BSR_HIST_BLOCKS_MAX,
			               img.tile_range, img.big_tiles, img.flags, &elems_sorted, &elems_free, &elems_compact, s);
		}
		STAGE_CHECK("binning", debug, s);
		{
			StageTimer t("sort_tiles", s);
			// (the 256 digit totals of pass 1 lie behind the rows of hist1)
			const uint32_t* digit_total1 = geom.hist1 + (size_t)256 * (((size_t)n_wg + 7) / 8 * 8);
			launch_sort_tiles(plan, T, (int)capacity, n_ptr, (int)capacity, img.tile_range, img.big_tiles, img.flags,
			                  digit_total1, elems_sorted, elems_free, bin.point_list, elems_compact,
			                  ((flags & BSR_FLAG_TEST_SORT_INT) ? 1 : 0) | ((flags & BSR_FLAG_TEST_SORT_NETWORK) ? 2 : 0),
			                  (flags & BSR_FLAG_TEST_SMALL_GRIDS) != 0, s);
		}
		STAGE_CHECK("sort_tiles", debug, s);
		{
			StageTimer t("render_fwd", s);
			// one view of at most 2^24 Gaussians: the forward's split-list staging hands its per-half box tests to the
			// backward in the top byte of the point_list words (flags[6] says whether it did)
			int* const masks_flag = (V == 1 && P <= (1 << 24) && !(flags & BSR_FLAG_TEST_NO_HALF_MASKS)) ? img.flags + 6 : nullptr;
			launch_render_fwd(gx, gy, V, width, height, n_ptr, (int)capacity, img.tile_range, bin.point_list, masks_flag, geom.rec,
			                  background, V > 1 ? nullptr : img.final_T, V > 1 ? nullptr : img.n_contrib, out_color, out_depth,
			                  (flags & BSR_FLAG_EXACT_EXP) != 0, no_readback, img.flags + BSR_POOL_FWD, s);
		}
		return 0;
	};
	if (no_readback) {
		// The caller's capacity sizes the scratch; nothing is waited for.  The counters still travel to the pinned buffer
		// (checked by this thread's next forward / bsr_check_deferred) unless the stream is capturing: an event recorded
		// into a graph cannot be waited for on the host.
		if (!capturing) {
			HIP_TRY(hipEventRecord(sc->deferred, s));   // (behind k_scans, which wrote the counters to the pinned buffer)
			sc->pending = true;
			sc->pending_capacity = (size_t)given_capacity;
			sc->pending_P = V == 1 ? P : -1; sc->pending_W = width; sc->pending_H = height;
		}
		// the binning plan: from the kept instances of this thread's last CHECKED no-readback forward of the same shape (its
		// counters were read when this call began), else from the capacity alone -- every plan is correct for every input
		const bool same = V == 1 && sc->nr_P == P && sc->nr_W == width && sc->nr_H == height;
		if (run_tail((size_t)given_capacity, false, same ? (long long)sc->nr_kept : 0)) return 1;
		STAGE_CHECK("render_fwd", debug, s);
		return 0;
	}
	HIP_TRY(hipEventRecord(sc->copied, s));   // behind k_scans: the counters are in the pinned buffer when it completes
	if (guess) {
		size_t c = (size_t)sc->last_R + (size_t)sc->last_R / 4 + 4096;
		if (c > 0x7fffffffu) c = 0x7fffffffu;
		if (run_tail(c, false, (long long)sc->last_kept)) return 1;   // the whole rest of the forward is in flight before the host waits
	}
	HIP_TRY(hipEventSynchronize(sc->copied));
	const int h_flag = prefiltered ? sc->pinned[0] : 0;
	const uint32_t h_kept = (uint32_t)sc->pinned[2], h_R = (uint32_t)sc->pinned[3];
	if (h_flag) return fail("Point is filtered although prefiltered is set. This shouldn't happen!");
	if (h_R > 0x7fffffffu) return fail("too many tile instances (%u)", h_R);
	const int R = (int)h_R;
	if (num_rendered) *num_rendered = R;
	// size hint for the next call: this call's count, but decaying only by 1/8 per call after a large view
	// (training visits views in random order; a short guess costs a second pass)
	{
		const bool same = sc->last_P == P && sc->last_W == width && sc->last_H == height && sc->last_V == V;
		const uint32_t decayed = same ? sc->last_R - sc->last_R / 8 : 0u;
		sc->last_P = P; sc->last_W = width; sc->last_H = height; sc->last_V = V;
		sc->last_R = h_R > decayed ? h_R : decayed;
		sc->last_kept = h_kept;
	}
	// The backward is handed R, not the capacity this call carved with: it finds point_list at the buffer's start and
	// puts its slab (up to 40 B per KEPT instance) right behind point_list[R].  A guessed buffer serves it only if that
	// R-based carve ends inside it -- with kept <= cap < R the slab could otherwise run past the end (cap + 512 Ki < R
	// is enough to get past the 2 MB of histogram space behind the work section).
	const bool backward_fits = V != 1 || align_up((size_t)R * sizeof(uint32_t), 256) + align_up((size_t)h_kept * BSR_SLAB_ROW_BYTES + BSR_SLAB_TAIL_BYTES, 256) <=
	                                         BinState::bytes(cap, true) - 256;
	if (!guess || (size_t)h_kept > cap || !backward_fits) {
		// first call of this shape, more kept instances than the guessed scratch holds (the kernels of the first
		// attempt then returned without touching anything), or a buffer the backward's carve would overrun
		if (run_tail((size_t)R, guess, (long long)h_kept)) return 1;
	}
	STAGE_CHECK("render_fwd", debug, s);
	return 0;
}


extern "C" {

int bsr_version(void) { return BSR_VERSION; }
const char* bsr_last_error(void) { return g_err; }

size_t bsr_geometry_bytes(int P) { return GeomState::bytes((size_t)(P > 0 ? P : 0)); }
size_t bsr_binning_bytes(int R) { return BinState::bytes((size_t)(R > 0 ? R : 0), true); }
size_t bsr_transmittance_offset(const void* image_buffer)
{
	return align_up((size_t)image_buffer, 256) - (size_t)image_buffer;   // ImgState::carve: final_T is the first section
}
size_t bsr_image_bytes(int W, int H)
{
	const size_t gx = (W + BSR_TILE - 1) / BSR_TILE, gy = (H + BSR_TILE - 1) / BSR_TILE;
	return ImgState::bytes((size_t)W * H, gx * gy);
}

int bsr_check_deferred(void)
{
	g_err[0] = 0;
	SyncCache* sc = sync_cache();
	if (!sc) return 1;
	return check_deferred(sc);
}

int bsr_read_counts(const char* image_buffer, int width, int height, void* stream, int* kept, int* num_rendered)
{
	g_err[0] = 0;
	if (!image_buffer || width <= 0 || height <= 0) return fail("bsr_read_counts: invalid image buffer / size");
	const size_t gx = (width + BSR_TILE - 1) / BSR_TILE, gy = (height + BSR_TILE - 1) / BSR_TILE;
	ImgState img = ImgState::carve(const_cast<char*>(image_buffer), (size_t)width * height, gx * gy);
	uint32_t k = 0, r = 0;
	if (read_u32_blocking((const uint32_t*)(img.flags + 2), &k, (hipStream_t)stream)) return 1;
	if (read_u32_blocking((const uint32_t*)(img.flags + 3), &r, (hipStream_t)stream)) return 1;
	if (kept) *kept = (int)k;
	if (num_rendered) *num_rendered = (int)r;
	return 0;
}

int bsr_profile_enable(int on)
{
	g_prof_on = on != 0;
	g_prof_every = on > 1 ? on : 1;
	return 0;
}

int bsr_profile_only(const char* stage)
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	g_prof_only[0] = 0;
	if (stage) {
		strncpy(g_prof_only, stage, sizeof(g_prof_only) - 1);
		g_prof_only[sizeof(g_prof_only) - 1] = 0;
	}
	return 0;
}

int bsr_profile_reset(void)
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	for (auto& r : g_stages) {
		drain_stage(r, true);
		r.total_ms = 0;
		r.launches = 0;
	}
	return 0;
}

int bsr_profile_read(bsr_stage_profile* out, int max_stages)
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	int n = 0;
	for (auto& r : g_stages) {
		drain_stage(r, true);
		if (n < max_stages) {
			out[n].name = r.name;
			out[n].total_ms = r.total_ms;
			out[n].launches = r.launches;
			n++;
		}
	}
	return n;
}

int bsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream)
{
	(void)projmatrix;
	g_err[0] = 0;
	hipStream_t s = (hipStream_t)stream;
	if (P <= 0) return 0;
	if (!means3D || !viewmatrix || !present) return fail("bsr_mark_visible: null pointer");
	{
		StageTimer t("mark_visible", s);
		launch_mark_visible(P, means3D, viewmatrix, present, s);
	}
	STAGE_CHECK("mark_visible", 0, s);
	return 0;
}

static int check_common(int P, int width, int height, const float* means3D, const float* scales, const float* rotations,
                        const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix)
{
	if (P < 0 || width <= 0 || height <= 0) return fail("invalid sizes: P=%d width=%d height=%d", P, width, height);
	if ((width + BSR_TILE - 1) / BSR_TILE > 65535 || (height + BSR_TILE - 1) / BSR_TILE > 65535)
		return fail("image too large for 16-bit tile coordinates");
	if (P > 0 && !means3D) return fail("means3D is null");
	if (!viewmatrix || !projmatrix) return fail("viewmatrix/projmatrix is null");
	const bool has_sr = scales != nullptr && rotations != nullptr;
	const bool has_any_sr = scales != nullptr || rotations != nullptr;
	if (P > 0 && ((!has_sr && !cov3D_precomp) || (has_any_sr && cov3D_precomp)))
		return fail("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
	return 0;
}

int bsr_visible_filter(int P, int M, int width, int height, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, float tan_fovx, float tan_fovy,
                       int prefiltered, int* radii, int debug, void* stream)
{
	(void)M;
	g_err[0] = 0;
	hipStream_t s = (hipStream_t)stream;
	if (check_common(P, width, height, means3D, scales, rotations, cov3D_precomp, viewmatrix, projmatrix)) return 1;
	if (P == 0) return 0;
	if (!radii) return fail("radii is null");
	// prefiltered: the kernel reports a culled point through ONE word of the calling thread's pinned HOST buffer (mapped
	// into the device's address space): no device allocation, nothing to free on an error path
	int* d_flag = nullptr;
	SyncCache* sc = nullptr;
	if (prefiltered) {
		sc = sync_cache();
		if (!sc) return 1;
		sc->pinned[5] = 0;
		d_flag = sc->pinned_dev + 5;
	}
	PreArgs a;
	memset(&a, 0, sizeof(a));
	a.P = P; a.D = 0; a.M = 0;
	a.means3D = means3D; a.scales = scales; a.scale_modifier = scale_modifier; a.rotations = rotations;
	a.cov3D_precomp = cov3D_precomp; a.viewmatrix = viewmatrix; a.projmatrix = projmatrix;
	a.W = width; a.H = height; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
	a.focal_y = height / (2.0f * tan_fovy);
	a.focal_x = width / (2.0f * tan_fovx);
	a.gx = (width + BSR_TILE - 1) / BSR_TILE; a.gy = (height + BSR_TILE - 1) / BSR_TILE;
	a.prefiltered = prefiltered; a.radii = radii; a.flags = d_flag;
	{
		StageTimer t("visible_filter", s);
		launch_preprocess(a, true, s);
	}
	STAGE_CHECK("visible_filter", debug, s);
	if (prefiltered) {
		HIP_TRY(hipStreamSynchronize(s));
		if (sc->pinned[5]) return fail("Point is filtered although prefiltered is set. This shouldn't happen!");
	}
	return 0;
}

int bsr_visible_filter_views(int P, int n_views, int width, int height, const float* means3D, const float* scales,
                             float scale_modifier, const float* rotations, const float* cov3D_precomp,
                             const float* viewmatrices, const float* projmatrices, float tan_fovx, float tan_fovy,
                             int* radii, int debug, void* stream)
{
	g_err[0] = 0;
	hipStream_t s = (hipStream_t)stream;
	if (n_views < 0) return fail("n_views must be >= 0");
	if (check_common(P, width, height, means3D, scales, rotations, cov3D_precomp, viewmatrices, projmatrices)) return 1;
	if (P == 0 || n_views == 0) return 0;
	if (!radii) return fail("radii is null");
	{
		StageTimer t("visible_filter_views", s);
		launch_visible_filter_views(P, n_views, means3D, scales, scale_modifier, rotations, cov3D_precomp, viewmatrices,
		                            projmatrices, width, height, tan_fovx, tan_fovy, radii, nullptr, 0, nullptr,
		                            nullptr, nullptr, s);
	}
	STAGE_CHECK("visible_filter_views", debug, s);
	return 0;
}

size_t bsr_visible_groups_scratch_bytes(int P, int n_groups)
{
	if (P <= 0 || n_groups <= 0) return 0;
	return sizeof(uint32_t) * (size_t)n_groups * (((size_t)P + 255) / 256);
}

int bsr_visible_filter_groups(int P, int n_views, int n_groups, int width, int height, const float* means3D,
                              const float* scales, float scale_modifier, const float* rotations,
                              const float* cov3D_precomp, const float* viewmatrices, const float* projmatrices,
                              float tan_fovx, float tan_fovy, const int* group_of_view, uint8_t* group_mask,
                              uint32_t* group_counts, void* count_scratch, int debug, void* stream)
{
	g_err[0] = 0;
	hipStream_t s = (hipStream_t)stream;
	if (n_views < 0 || n_groups < 0 || n_groups > 64) return fail("bsr_visible_filter_groups: need 0 <= n_groups <= 64");
	if (check_common(P, width, height, means3D, scales, rotations, cov3D_precomp, viewmatrices, projmatrices)) return 1;
	if (P == 0 || n_groups == 0) {   // (P == 0: group_mask has no elements; n_groups == 0: no rows)
		if (group_counts && n_groups > 0) HIP_TRY(hipMemsetAsync(group_counts, 0, sizeof(uint32_t) * (size_t)n_groups, s));
		return 0;
	}
	if (!group_mask || (n_views > 0 && !group_of_view)) return fail("bsr_visible_filter_groups: NULL buffer");
	if (group_counts && !count_scratch)
		return fail("bsr_visible_filter_groups: group_counts needs count_scratch (bsr_visible_groups_scratch_bytes)");
	{
		StageTimer t("visible_filter_groups", s);
		// per-workgroup partial counts: the caller's scratch, like every other byte of device memory this library uses
		uint32_t* wg_counts = group_counts ? (uint32_t*)count_scratch : nullptr;
		launch_visible_filter_views(P, n_views, means3D, scales, scale_modifier, rotations, cov3D_precomp, viewmatrices,
		                            projmatrices, width, height, tan_fovx, tan_fovy, nullptr, group_of_view, n_groups,
		                            group_mask, wg_counts, group_counts, s);
	}
	STAGE_CHECK("visible_filter_groups", debug, s);
	return 0;
}

static int gather_impl(const char* who, int R, int P, int n_src, const float* const* src, const int* widths,
                       const int64_t* idx, int idx_stride, float* dst_packed, float* const* dst_each, int debug,
                       void* stream)
{
	g_err[0] = 0;
	hipStream_t s = (hipStream_t)stream;
	if (R < 0 || P < 0 || n_src < 1 || n_src > BSR_PACK_MAX_SRC) return fail("%s: need R >= 0, P >= 0 and 1 <= n_src <= 8", who);
	if (!src || !widths) return fail("%s: NULL table", who);
	int row = 0;
	for (int k = 0; k < n_src; k++) {
		if (widths[k] < 1 || !src[k]) return fail("%s: every source needs a pointer and a width >= 1", who);
		row += widths[k];
	}
	if (row > 4096) return fail("%s: more than 4096 floats per row", who);
	if (R == 0) return 0;
	if (!idx || idx_stride < 1 || (!dst_packed && !dst_each)) return fail("%s: NULL buffer", who);
	if (!dst_packed)
		for (int k = 0; k < n_src; k++)
			if (!dst_each[k]) return fail("%s: NULL destination", who);
	{
		StageTimer t(who, s);
		launch_pack_rows(R, P, n_src, src, widths, idx, idx_stride, dst_packed, dst_each, s);
	}
	STAGE_CHECK(who, debug, s);
	return 0;
}

int bsr_pack_rows(int R, int P, int n_src, const float* const* src, const int* widths, const int64_t* idx, int idx_stride,
                  float* dst, int debug, void* stream)
{
	return gather_impl("pack_rows", R, P, n_src, src, widths, idx, idx_stride, dst, nullptr, debug, stream);
}

int bsr_gather_rows(int R, int P, int n_src, const float* const* src, const int* widths, const int64_t* idx,
                    int idx_stride, float* const* dst, int debug, void* stream)
{
	return gather_impl("gather_rows", R, P, n_src, src, widths, idx, idx_stride, nullptr, dst, debug, stream);
}

int bsr_forward(bsr_alloc_fn geometryBuffer, void* geometry_user, bsr_alloc_fn binningBuffer, void* binning_user,
                bsr_alloc_fn imageBuffer, void* image_user, int P, int D, int M, const float* background, int width,
                int height, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii,
                int debug, void* stream, int* num_rendered)
{
	return forward_impl(1, geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, P, D, M,
	                    background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
	                    rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
	                    out_color, out_depth, radii, debug, stream, num_rendered, 0u);
}

int bsr_forward_ex(bsr_alloc_fn geometryBuffer, void* geometry_user, bsr_alloc_fn binningBuffer, void* binning_user,
                   bsr_alloc_fn imageBuffer, void* image_user, int P, int D, int M, const float* background, int width,
                   int height, const float* means3D, const float* shs, const float* colors_precomp,
                   const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                   const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                   float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii,
                   int debug, void* stream, int* num_rendered, unsigned flags)
{
	return forward_impl(1, geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, P, D, M,
	                    background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
	                    rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
	                    out_color, out_depth, radii, debug, stream, num_rendered, flags);
}

int bsr_forward_views(bsr_alloc_fn geometryBuffer, void* geometry_user, bsr_alloc_fn binningBuffer, void* binning_user,
                      bsr_alloc_fn imageBuffer, void* image_user, int P, int D, int M, int n_views,
                      const float* background, int width, int height, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                      const float* rotations, const float* cov3D_precomp, const float* viewmatrices,
                      const float* projmatrices, const float* cam_positions, float tan_fovx, float tan_fovy,
                      int prefiltered, float* out_color, float* out_depth, int* radii, int debug, void* stream,
                      int* num_rendered, unsigned flags)
{
	g_err[0] = 0;
	if (num_rendered) *num_rendered = 0;
	if (n_views < 0) return fail("n_views must be >= 0");
	if (n_views == 0) return 0;
	return forward_impl(n_views, geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, P, D,
	                    M, background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
	                    rotations, cov3D_precomp, viewmatrices, projmatrices, cam_positions, tan_fovx, tan_fovy,
	                    prefiltered, out_color, out_depth, radii, debug, stream, num_rendered, flags);
}

}  // extern "C"

// out_depth == nullptr: the reference's backward (dL_depths ignored); otherwise the depth-gradient extension
static int backward_impl(int P, int D, int M, int R, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* scales, float scale_modifier,
                         const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                         const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                         char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                         const float* out_depth, const float* dL_depths, float* dL_dmean2D, float* dL_dconic,
                         float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                         float* dL_dscale, float* dL_drot, int debug, void* stream, unsigned flags)
{
	g_err[0] = 0;
	hipStream_t s = (hipStream_t)stream;
	// (the forward's flags are accepted and ignored, so that a caller can hand one word to both calls)
	if (flags & ~(unsigned)(BSR_FLAG_EXACT_GRAD | BSR_FLAG_EXACT_EXP | BSR_FLAG_NO_READBACK | BSR_FLAG_TEST_MASK))
		return fail("backward: unknown flag bits 0x%x",
		            flags & ~(unsigned)(BSR_FLAG_EXACT_GRAD | BSR_FLAG_EXACT_EXP | BSR_FLAG_NO_READBACK | BSR_FLAG_TEST_MASK));
	if (P == 0) return 0;
	if (check_common(P, width, height, means3D, scales, rotations, cov3D_precomp, viewmatrix, projmatrix)) return 1;
	if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer)) return fail("scratch buffer is null");
	if (!dL_dpix || !background) return fail("dL_dpix/background is null");
	if (!dL_dmean2D || !dL_dopacity || !dL_dmean3D) return fail("gradient output is null");
	// intermediate results may be declined (NULL) -- unless they are the gradient of an input that was given
	if (!shs && !dL_dcolor) return fail("dL_dcolor is null but colors_precomp is the colour input");
	if (!scales && !dL_dcov3D) return fail("dL_dcov3D is null but cov3D_precomp is the covariance input");
	if (shs && (M <= 0 || !dL_dsh || !campos)) return fail("SH backward needs M > 0, dL_dsh and campos");
	if (shs && (D < 0 || (D + 1) * (D + 1) > M))
		return fail("sh_degree %d needs %d coefficients per Gaussian, shs holds %d", D, (D + 1) * (D + 1), M);
	if (scales && (!dL_dscale || !dL_drot)) return fail("dL_dscale/dL_drot is null");

	const int gx = (width + BSR_TILE - 1) / BSR_TILE, gy = (height + BSR_TILE - 1) / BSR_TILE;
	const int T = gx * gy;
	const size_t N = (size_t)width * height;
	GeomState geom = GeomState::carve(geom_buffer, (size_t)P);
	ImgState img = ImgState::carve(image_buffer, N, (size_t)T);
	BinState bin = BinState::carve(binning_buffer, (size_t)(R > 0 ? R : 0), true);

	// slab[R][9 or 10] f32 (tight rows): per-instance partial sums, Gaussian-major (kept instances only use the first R_kept rows), in the
	// caller's binning buffer over the forward's dead radix ping-pong buffers.  The forward sized the buffer for R or for
	// a guessed capacity it checked against this very carve (forward_impl: backward_fits): point_list[R], then the rows
	// of the kept instances, end inside it.
	float4* slab = bin.slab;

	if (R > 0) {
		{
			StageTimer t("render_bwd", s);
			launch_render_bwd(gx, gy, width, height, img.tile_range, bin.point_list, geom.rec, geom.wg_kept, background, img.final_T,
			                  img.n_contrib, dL_dpix, out_depth, out_depth ? dL_depths : nullptr, img.flags + 6, slab,
			                  (flags & BSR_FLAG_EXACT_GRAD) != 0, R, s);
		}
		STAGE_CHECK("render_bwd", debug, s);
	}
	BwdArgs a;
	memset(&a, 0, sizeof(a));
	a.P = P; a.D = D; a.M = M;
	a.means3D = means3D; a.radii = radii; a.shs = shs; a.scales = scales; a.rotations = rotations;
	a.scale_modifier = scale_modifier; a.cov3D_precomp = cov3D_precomp; a.viewmatrix = viewmatrix;
	a.projmatrix = projmatrix; a.campos = campos; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
	a.focal_y = height / (2.0f * tan_fovy);
	a.focal_x = width / (2.0f * tan_fovx);
	a.geom = geom;
	a.slab = slab; a.depth_grad = out_depth != nullptr;
	a.kept_ptr = img.flags + 2; a.capacity = R;
	a.dL_dmean2D = dL_dmean2D; a.dL_dconic = dL_dconic; a.dL_dopacity = dL_dopacity; a.dL_dcolor = dL_dcolor;
	a.dL_dmean3D = dL_dmean3D; a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscale = dL_dscale; a.dL_drot = dL_drot;
	{
		StageTimer t("preprocess_bwd", s);
		launch_preprocess_bwd(a, s);
	}
	STAGE_CHECK("preprocess_bwd", debug, s);
	return 0;
}

extern "C" {

int bsr_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer,
                 char* binning_buffer, char* image_buffer, const float* dL_dpix, const float* dL_depths,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                 float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream)
{
	(void)dL_depths;   // accepted and ignored, as in the reference (backward.cu:457-463,539-554)
	(void)colors_precomp;
	return backward_impl(P, D, M, R, background, width, height, means3D, shs, scales, scale_modifier, rotations,
	                     cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer,
	                     binning_buffer, image_buffer, dL_dpix, nullptr, nullptr, dL_dmean2D, dL_dconic, dL_dopacity,
	                     dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream, 0u);
}

int bsr_backward_depth(int P, int D, int M, int R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                       float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                       const float* out_depth, const float* dL_dpix, const float* dL_depths, float* dL_dmean2D,
                       float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                       float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream)
{
	(void)colors_precomp;
	if (P > 0 && (!out_depth || !dL_depths)) {
		g_err[0] = 0;
		return fail("bsr_backward_depth needs out_depth and dL_depths");
	}
	return backward_impl(P, D, M, R, background, width, height, means3D, shs, scales, scale_modifier, rotations,
	                     cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer,
	                     binning_buffer, image_buffer, dL_dpix, out_depth, dL_depths, dL_dmean2D, dL_dconic, dL_dopacity,
	                     dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream, 0u);
}

int bsr_backward_ex(int P, int D, int M, int R, const float* background, int width, int height,
                    const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                    float scale_modifier, const float* rotations, const float* cov3D_precomp,
                    const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                    float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                    const float* out_depth, const float* dL_dpix, const float* dL_depths, float* dL_dmean2D,
                    float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                    float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream, unsigned flags)
{
	(void)colors_precomp;
	if (P > 0 && out_depth && !dL_depths) {
		g_err[0] = 0;
		return fail("bsr_backward_ex: out_depth selects the depth-gradient extension, which needs dL_depths");
	}
	return backward_impl(P, D, M, R, background, width, height, means3D, shs, scales, scale_modifier, rotations,
	                     cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer,
	                     binning_buffer, image_buffer, dL_dpix, out_depth, out_depth ? dL_depths : nullptr, dL_dmean2D,
	                     dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream,
	                     flags);
}

}  // extern "C"
