// A consumer of the C ABI that knows nothing about torch or python: plain hipMalloc'd buffers, C callbacks for
// the three scratch buffers, bsr_forward + bsr_backward (+ the camera-sweep helpers: group filter, row gathers) through include/bloomscene_rast.h.  This is the shape
// of the binding a maintainer of the reference would write in rasterize_points.cu (option B).
// tests/test_abi_native_gpu.py feeds it a seeded scene and compares what it writes with the CPU oracle.
//
//   abi_roundtrip <inputs.bin> <outputs.bin>
//
// inputs.bin : int32 P, D, M, W, H, use_sh; float tan_fovx, tan_fovy, scale_modifier; then float arrays
//              bg[3] means3D[3P] colour[use_sh ? 3MP : 3P] opacities[P] scales[3P] rotations[4P] view[16] proj[16]
//              campos[3] dL_dcolor[3HW] dL_ddepth[HW]
// outputs.bin: int32 num_rendered; float color[3HW] depth[HW]; int32 radii[P]; float dL_dmean3D[3P] dL_dmean2D[3P]
//              dL_dopacity[P] dL_dcolour[3P] dL_dsh[3MP if use_sh] dL_dscale[3P] dL_drot[4P]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "bloomscene_rast.h"

#define HIP_OK(x)                                                                      \
	do {                                                                               \
		hipError_t e_ = (x);                                                           \
		if (e_ != hipSuccess) {                                                        \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
			return 2;                                                                  \
		}                                                                              \
	} while (0)

struct Scratch {   // what resizeFunctional(torch::Tensor&) is in the reference: a growable device buffer
	char* ptr = nullptr;
	size_t cap = 0;
	int calls = 0;
};
static char* grow(void* user, size_t bytes)
{
	Scratch* s = static_cast<Scratch*>(user);
	s->calls++;
	if (bytes > s->cap) {
		if (s->ptr) (void)hipFree(s->ptr);
		if (hipMalloc((void**)&s->ptr, bytes ? bytes : 1) != hipSuccess) return nullptr;
		s->cap = bytes;
	}
	return s->ptr;
}

template <typename T>
static bool read_vec(FILE* f, std::vector<T>& v, size_t n)
{
	v.resize(n);
	return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}
template <typename T>
static T* to_device(const std::vector<T>& v)
{
	T* d = nullptr;
	if (hipMalloc((void**)&d, v.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
	if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
	return d;
}
template <typename T>
static T* device_out(size_t n)
{
	T* d = nullptr;
	if (hipMalloc((void**)&d, n * sizeof(T) + 16) != hipSuccess) return nullptr;
	(void)hipMemset(d, 0xff, n * sizeof(T));   // NaN / -1 patterns: the library must overwrite everything
	return d;
}
template <typename T>
static bool write_from_device(FILE* f, const T* d, size_t n)
{
	std::vector<T> h(n);
	if (n && hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return false;
	return n == 0 || fwrite(h.data(), sizeof(T), n, f) == n;
}

int main(int argc, char** argv)
{
	if (argc != 3 && argc != 4) {
		fprintf(stderr, "usage: %s inputs.bin outputs.bin [exact|strict|nowait]\n", argv[0]);
		return 2;
	}
	// Numerics are per call.  No third argument: the reference-shaped entry points bsr_forward / bsr_backward (the
	// library's default).  "exact": bsr_forward_ex / bsr_backward_ex with BSR_FLAG_EXACT_EXP (the pinned exp on every
	// evaluation of the forward blend: bit-for-bit comparable with the CPU oracle); "strict": BSR_FLAG_EXACT_GRAD as well
	// (the reference's per-pair operations in the backward tile walk).
	const bool use_ex = argc == 4;
	unsigned flags = 0;
	if (use_ex) {
		if (!strcmp(argv[3], "exact")) flags = BSR_FLAG_EXACT_EXP;
		else if (!strcmp(argv[3], "strict")) flags = BSR_FLAG_EXACT_EXP | BSR_FLAG_EXACT_GRAD;
		// "nowait": exact numerics through BSR_FLAG_NO_READBACK -- the forward is handed a capacity instead of waiting for
		// num_rendered (a first, ordinary call tells it), overflow reporting and bsr_read_counts are checked on the way
		else if (!strcmp(argv[3], "nowait")) flags = BSR_FLAG_EXACT_EXP | BSR_FLAG_NO_READBACK;
		else { fprintf(stderr, "unknown mode %s\n", argv[3]); return 2; }
	}
	FILE* fi = fopen(argv[1], "rb");
	if (!fi) { perror(argv[1]); return 2; }
	int32_t hdr[6];
	float fl[3];
	if (fread(hdr, 4, 6, fi) != 6 || fread(fl, 4, 3, fi) != 3) { fprintf(stderr, "short header\n"); return 2; }
	const int P = hdr[0], D = hdr[1], M = hdr[2], W = hdr[3], H = hdr[4], use_sh = hdr[5];
	const size_t N = (size_t)W * H;
	std::vector<float> bg, means, colour, opac, scales, rots, view, proj, campos, gC, gD;
	bool ok = read_vec(fi, bg, 3) && read_vec(fi, means, (size_t)3 * P) &&
	          read_vec(fi, colour, use_sh ? (size_t)3 * M * P : (size_t)3 * P) && read_vec(fi, opac, (size_t)P) &&
	          read_vec(fi, scales, (size_t)3 * P) && read_vec(fi, rots, (size_t)4 * P) && read_vec(fi, view, 16) &&
	          read_vec(fi, proj, 16) && read_vec(fi, campos, 3) && read_vec(fi, gC, 3 * N) && read_vec(fi, gD, N);
	fclose(fi);
	if (!ok) { fprintf(stderr, "short input file\n"); return 2; }
	if (bsr_version() != BSR_VERSION) { fprintf(stderr, "header / library version mismatch\n"); return 2; }

	hipStream_t stream;
	HIP_OK(hipStreamCreate(&stream));   // deliberately not the null stream
	float *d_bg = to_device(bg), *d_means = to_device(means), *d_col = to_device(colour), *d_op = to_device(opac),
	      *d_sc = to_device(scales), *d_rot = to_device(rots), *d_view = to_device(view), *d_proj = to_device(proj),
	      *d_cam = to_device(campos), *d_gC = to_device(gC), *d_gD = to_device(gD);
	float* out_color = device_out<float>(3 * N);
	float* out_depth = device_out<float>(N);
	int* radii = device_out<int>((size_t)P);
	if (!d_bg || !d_means || !d_col || !d_op || !d_sc || !d_rot || !d_view || !d_proj || !d_cam || !d_gC || !d_gD ||
	    !out_color || !out_depth || !radii) {
		fprintf(stderr, "device allocation failed\n");
		return 2;
	}
	Scratch geom, binning, img;
	int num_rendered = -1;
	int true_num_rendered = -1;
	if (flags & BSR_FLAG_NO_READBACK) {
		// an ordinary forward first: its counts size the capacity.  Then a capacity that is too small: NaN frame, an
		// error from the NEXT forward of this thread, the library usable afterwards.
		int r0 = -1, kept = -1, r1 = -1;
		if (bsr_forward_ex(grow, &geom, grow, &binning, grow, &img, P, D, use_sh ? M : 0, d_bg, W, H, d_means,
		                   use_sh ? d_col : nullptr, use_sh ? nullptr : d_col, d_op, d_sc, fl[2], d_rot, nullptr, d_view, d_proj,
		                   d_cam, fl[0], fl[1], 0, out_color, out_depth, radii, 0, stream, &r0, BSR_FLAG_EXACT_EXP) != 0) {
			fprintf(stderr, "probe forward: %s\n", bsr_last_error());
			return 1;
		}
		if (bsr_read_counts(img.ptr, W, H, stream, &kept, &r1) != 0 || r1 != r0 || kept <= 0 || kept > r0) {
			fprintf(stderr, "bsr_read_counts: kept %d num_rendered %d vs %d (%s)\n", kept, r1, r0, bsr_last_error());
			return 1;
		}
		true_num_rendered = r0;
		int small = kept / 2 > 0 ? kept / 2 : 1;
		if (bsr_forward_ex(grow, &geom, grow, &binning, grow, &img, P, D, use_sh ? M : 0, d_bg, W, H, d_means,
		                   use_sh ? d_col : nullptr, use_sh ? nullptr : d_col, d_op, d_sc, fl[2], d_rot, nullptr, d_view, d_proj,
		                   d_cam, fl[0], fl[1], 0, out_color, out_depth, radii, 0, stream, &small, flags) != 0) {
			fprintf(stderr, "overflowing forward must be accepted: %s\n", bsr_last_error());
			return 1;
		}
		float probe[2] = {0.f, 0.f};
		HIP_OK(hipMemcpyAsync(probe, out_color, sizeof(float), hipMemcpyDeviceToHost, stream));
		HIP_OK(hipMemcpyAsync(probe + 1, out_depth + N / 2, sizeof(float), hipMemcpyDeviceToHost, stream));
		HIP_OK(hipStreamSynchronize(stream));
		if (probe[0] == probe[0] || probe[1] == probe[1]) { fprintf(stderr, "overflowed frame is not NaN\n"); return 1; }
		if (bsr_check_deferred() == 0 || !strstr(bsr_last_error(), "was not rendered")) {
			fprintf(stderr, "overflow not reported: '%s'\n", bsr_last_error());
			return 1;
		}
		if (bsr_check_deferred() != 0) { fprintf(stderr, "the deferred error must be reported once\n"); return 1; }
		num_rendered = r0 + r0 / 4 + 64;   // the capacity of the call below (on entry), and what the backward is handed
		printf("nowait: kept %d of %d instances; overflow at capacity %d reported\n", kept, r0, small);
	}
	int rc = use_ex
	    ? bsr_forward_ex(grow, &geom, grow, &binning, grow, &img, P, D, use_sh ? M : 0, d_bg, W, H, d_means,
	                     use_sh ? d_col : nullptr, use_sh ? nullptr : d_col, d_op, d_sc, fl[2], d_rot, nullptr, d_view,
	                     d_proj, d_cam, fl[0], fl[1], 0, out_color, out_depth, radii, 0, stream, &num_rendered, flags)
	    : bsr_forward(grow, &geom, grow, &binning, grow, &img, P, D, use_sh ? M : 0, d_bg, W, H, d_means,
	                  use_sh ? d_col : nullptr, use_sh ? nullptr : d_col, d_op, d_sc, fl[2], d_rot, nullptr, d_view,
	                  d_proj, d_cam, fl[0], fl[1], 0, out_color, out_depth, radii, 0, stream, &num_rendered);
	if (rc != 0) { fprintf(stderr, "bsr_forward: %s\n", bsr_last_error()); return 1; }

	float* g_mean2D = device_out<float>((size_t)3 * P);
	float* g_conic = device_out<float>((size_t)4 * P);
	float* g_opac = device_out<float>((size_t)P);
	float* g_col = device_out<float>((size_t)3 * P);
	float* g_mean3D = device_out<float>((size_t)3 * P);
	float* g_cov3D = device_out<float>((size_t)6 * P);
	float* g_sh = device_out<float>(use_sh ? (size_t)3 * M * P : 1);
	float* g_scale = device_out<float>((size_t)3 * P);
	float* g_rot = device_out<float>((size_t)4 * P);
	rc = use_ex
	    ? bsr_backward_ex(P, D, use_sh ? M : 0, num_rendered, d_bg, W, H, d_means, use_sh ? d_col : nullptr,
	                      use_sh ? nullptr : d_col, d_sc, fl[2], d_rot, nullptr, d_view, d_proj, d_cam, fl[0], fl[1], radii,
	                      geom.ptr, binning.ptr, img.ptr, /*out_depth: the reference's backward*/ nullptr, d_gC, d_gD,
	                      g_mean2D, g_conic, g_opac, g_col, g_mean3D, g_cov3D, use_sh ? g_sh : nullptr, g_scale, g_rot, 0,
	                      stream, flags)
	    : bsr_backward(P, D, use_sh ? M : 0, num_rendered, d_bg, W, H, d_means, use_sh ? d_col : nullptr,
	                   use_sh ? nullptr : d_col, d_sc, fl[2], d_rot, nullptr, d_view, d_proj, d_cam, fl[0], fl[1], radii,
	                   geom.ptr, binning.ptr, img.ptr, d_gC, d_gD, g_mean2D, g_conic, g_opac, g_col, g_mean3D, g_cov3D,
	                   use_sh ? g_sh : nullptr, g_scale, g_rot, 0, stream);
	if (rc != 0) { fprintf(stderr, "bsr_backward: %s\n", bsr_last_error()); return 1; }
	HIP_OK(hipStreamSynchronize(stream));

	// ---- the camera-sweep helpers through the same header: group filter (this one camera as a path of one view in one
	// group) + its row count, then the rows it kept gathered both ways; checked here against the forward's radii and a
	// gather on the host
	{
		std::vector<int> h_radii((size_t)P);
		HIP_OK(hipMemcpy(h_radii.data(), radii, (size_t)P * sizeof(int), hipMemcpyDeviceToHost));
		int group0 = 0;
		int* d_group = nullptr;
		uint8_t* d_mask = nullptr;
		uint32_t* d_count = nullptr;
		HIP_OK(hipMalloc((void**)&d_group, sizeof(int)));
		HIP_OK(hipMemcpy(d_group, &group0, sizeof(int), hipMemcpyHostToDevice));
		HIP_OK(hipMalloc((void**)&d_mask, (size_t)P + 16));
		HIP_OK(hipMalloc((void**)&d_count, sizeof(uint32_t)));
		void* d_count_scratch = nullptr;   // the filter's per-workgroup counts: the caller's memory, sized by the library
		HIP_OK(hipMalloc(&d_count_scratch, bsr_visible_groups_scratch_bytes(P, 1) + 16));
		rc = bsr_visible_filter_groups(P, 1, 1, W, H, d_means, d_sc, fl[2], d_rot, nullptr, d_view, d_proj, fl[0], fl[1],
		                               d_group, d_mask, d_count, d_count_scratch, 0, stream);
		if (rc != 0) { fprintf(stderr, "bsr_visible_filter_groups: %s\n", bsr_last_error()); return 1; }
		HIP_OK(hipStreamSynchronize(stream));
		std::vector<uint8_t> h_mask((size_t)P);
		uint32_t h_count = 0;
		HIP_OK(hipMemcpy(h_mask.data(), d_mask, (size_t)P, hipMemcpyDeviceToHost));
		HIP_OK(hipMemcpy(&h_count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
		std::vector<int64_t> idx;
		for (int i = 0; i < P; i++) {
			if ((h_mask[(size_t)i] != 0) != (h_radii[(size_t)i] > 0)) { fprintf(stderr, "group mask != radii > 0 at %d\n", i); return 1; }
			if (h_mask[(size_t)i]) idx.push_back(i);
		}
		if (h_count != idx.size()) { fprintf(stderr, "group count %u != %zu\n", h_count, idx.size()); return 1; }
		const int R = (int)idx.size();
		if (R > 0) {
			int64_t* d_idx = to_device(idx);
			const float* srcs[3] = {d_means, d_op, d_rot};
			const int widths[3] = {3, 1, 4};
			float* packed = device_out<float>((size_t)R * 8);
			float* each[3] = {device_out<float>((size_t)R * 3), device_out<float>((size_t)R), device_out<float>((size_t)R * 4)};
			if (!d_idx || !packed || !each[0] || !each[1] || !each[2]) { fprintf(stderr, "device allocation failed\n"); return 2; }
			rc = bsr_pack_rows(R, P, 3, srcs, widths, d_idx, 1, packed, 0, stream);
			if (rc != 0) { fprintf(stderr, "bsr_pack_rows: %s\n", bsr_last_error()); return 1; }
			rc = bsr_gather_rows(R, P, 3, srcs, widths, d_idx, 1, each, 0, stream);
			if (rc != 0) { fprintf(stderr, "bsr_gather_rows: %s\n", bsr_last_error()); return 1; }
			HIP_OK(hipStreamSynchronize(stream));
			std::vector<float> h_packed((size_t)R * 8), h0((size_t)R * 3), h1((size_t)R), h2((size_t)R * 4);
			HIP_OK(hipMemcpy(h_packed.data(), packed, h_packed.size() * 4, hipMemcpyDeviceToHost));
			HIP_OK(hipMemcpy(h0.data(), each[0], h0.size() * 4, hipMemcpyDeviceToHost));
			HIP_OK(hipMemcpy(h1.data(), each[1], h1.size() * 4, hipMemcpyDeviceToHost));
			HIP_OK(hipMemcpy(h2.data(), each[2], h2.size() * 4, hipMemcpyDeviceToHost));
			for (int r = 0; r < R; r++) {
				const size_t i = (size_t)idx[(size_t)r];
				float want[8] = {means[3 * i], means[3 * i + 1], means[3 * i + 2], opac[i],
				                 rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]};
				if (memcmp(want, &h_packed[(size_t)r * 8], sizeof(want)) != 0 || memcmp(want, &h0[(size_t)r * 3], 12) != 0 ||
				    memcmp(want + 3, &h1[(size_t)r], 4) != 0 || memcmp(want + 4, &h2[(size_t)r * 4], 16) != 0) {
					fprintf(stderr, "gathered row %d differs\n", r);
					return 1;
				}
			}
		}
		printf("sweep helpers ok: %u of %d rows kept by the group filter, packed and gathered\n", h_count, P);
	}

	FILE* fo = fopen(argv[2], "wb");
	if (!fo) { perror(argv[2]); return 2; }
	if (flags & BSR_FLAG_NO_READBACK) {
		if (bsr_check_deferred() != 0) { fprintf(stderr, "deferred: %s\n", bsr_last_error()); return 1; }
		num_rendered = true_num_rendered;   // (the file carries the frame's real count, as in the other modes)
	}
	int32_t nr = num_rendered;
	ok = fwrite(&nr, 4, 1, fo) == 1 && write_from_device(fo, out_color, 3 * N) && write_from_device(fo, out_depth, N) &&
	     write_from_device(fo, radii, (size_t)P) && write_from_device(fo, g_mean3D, (size_t)3 * P) &&
	     write_from_device(fo, g_mean2D, (size_t)3 * P) && write_from_device(fo, g_opac, (size_t)P) &&
	     write_from_device(fo, g_col, (size_t)3 * P) && write_from_device(fo, g_sh, use_sh ? (size_t)3 * M * P : 0) &&
	     write_from_device(fo, g_scale, (size_t)3 * P) && write_from_device(fo, g_rot, (size_t)4 * P);
	fclose(fo);
	if (!ok) { fprintf(stderr, "writing outputs failed\n"); return 2; }
	printf("abi_roundtrip ok: P=%d num_rendered=%d scratch callbacks geom/binning/image = %d/%d/%d\n", P, num_rendered,
	       geom.calls, binning.calls, img.calls);
	return 0;
}
