/*
 * bsr_oracle.c -- CPU ORACLE for the BloomScene depth-diff Gaussian rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported CPU baseline.  The product path (bloomscene_amd/) never calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 * and cannot be compiled here (needs nvcc, CUB and the un-vendored GLM submodule), so this
 * restatement is pinned only by (a) line-by-line citation of the reference sources below,
 * (b) an independent PyTorch-autograd cross-check (oracle/torch_splat.py) and (c) the
 * invariants in tests/.  See DESIGN.md "Oracle".
 *
 * Every function cites the reference file:line it restates.  Shorthand:
 *   CR/  = submodules/depth-diff-gaussian-rasterization/cuda_rasterizer/
 *   RAST/= submodules/depth-diff-gaussian-rasterization/
 *
 * Third-party arithmetic that is absent from /root/reference and restated here:
 *   - GLM (g-truc/glm 0.9.9.x, un-pinned submodule RAST/third_party/glm): mat3 is
 *     column-major, M[c][r] is column c row r, mat3(a,b,c,...) fills columns,
 *     (A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2] (that order),
 *     dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z, length = sqrt(dot), vec/scalar divides
 *     per component.
 *   - CUB (CUDA 11.7): DeviceScan::InclusiveSum and the STABLE LSD
 *     DeviceRadixSort::SortPairs on a bit range.
 *   - CUDA libdevice expf (<= 2 ulp, not correctly rounded): replaced by bsro_expf below,
 *     a <= 1 ulp Cody-Waite/Taylor exp built only from IEEE fma/mul/rint/ldexp so the HIP
 *     path can evaluate the *same* function and be compared bit for bit.
 *
 * All arithmetic is IEEE binary32 evaluated in source order (build with
 * -ffp-contract=off, no fast-math); the only fused operations are the explicit fmaf() calls of
 * bsro_expf and of the forward blend accumulations (bsro_render_forward, where the reference's
 * nvcc build contracts `+= a * b * c` into one FMA).  The unordered float atomicAdd sums of the reference
 * backward are accumulated in binary64 in a fixed order and rounded once.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16 /* CR/config.h:15 */
#define BLOCK_Y 16 /* CR/config.h:16 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y) /* CR/auxiliary.h:18 */
#define NUM_CHANNELS 3 /* CR/config.h:14 */

/* CR/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* ---------------------------------------------------------------- pinned exp */
/* Stand-in for CUDA's exp(float) at CR/forward.cu:426 and CR/backward.cu:512. */
float bsro_expf(float x)
{
	if (x != x) return x;
	if (x < -104.0f) return 0.0f;
	if (x > 88.72284f) return INFINITY;
	const float k = rintf(x * 1.44269504088896341f);
	float r = fmaf(-k, 0.693359375f, x);
	r = fmaf(-k, -2.12194440e-4f, r);
	float p = 1.0f / 5040.0f;
	p = fmaf(p, r, 1.0f / 720.0f);
	p = fmaf(p, r, 1.0f / 120.0f);
	p = fmaf(p, r, 1.0f / 24.0f);
	p = fmaf(p, r, 1.0f / 6.0f);
	p = fmaf(p, r, 0.5f);
	p = fmaf(p, r, 1.0f);
	p = fmaf(p, r, 1.0f);
	return ldexpf(p, (int)k);
}

void bsro_expf_array(int n, const float* x, float* y)
{
	for (int i = 0; i < n; i++) y[i] = bsro_expf(x[i]);
}

/* ---------------------------------------------------------------- GLM stand-ins */
typedef struct { float x, y, z; } vec3;
typedef struct { float c[3][3]; } mat3; /* c[col][row] */

static mat3 m3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2)
{
	mat3 m = {{{a0, a1, a2}, {b0, b1, b2}, {c0, c1, c2}}};
	return m;
}
static mat3 m3_mul(mat3 A, mat3 B)
{
	mat3 R;
	for (int c = 0; c < 3; c++)
		for (int r = 0; r < 3; r++)
			R.c[c][r] = A.c[0][r] * B.c[c][0] + A.c[1][r] * B.c[c][1] + A.c[2][r] * B.c[c][2];
	return R;
}
static mat3 m3_t(mat3 A)
{
	mat3 R;
	for (int c = 0; c < 3; c++)
		for (int r = 0; r < 3; r++) R.c[c][r] = A.c[r][c];
	return R;
}
static float dot3(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static float col_dot(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* ---------------------------------------------------------------- CR/auxiliary.h helpers */
/* CR/auxiliary.h:41-44: double-precision literals => evaluated in double. */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* CR/auxiliary.h:46-56 */
static void getRect(float px, float py, int max_radius, int gx, int gy, int* rmin, int* rmax)
{
	rmin[0] = imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
	rmin[1] = imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
	rmax[0] = imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
	rmax[1] = imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* CR/auxiliary.h:58-66 */
static vec3 transformPoint4x3(vec3 p, const float* m)
{
	vec3 t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
	          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
	return t;
}
/* CR/auxiliary.h:68-77 */
static void transformPoint4x4(vec3 p, const float* m, float* o)
{
	o[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
	o[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
	o[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
	o[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}
/* CR/auxiliary.h:89-97 */
static vec3 transformVec4x3Transpose(vec3 p, const float* m)
{
	vec3 t = {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
	          m[8] * p.x + m[9] * p.y + m[10] * p.z};
	return t;
}
/* CR/auxiliary.h:107-117 */
static vec3 dnormvdv(vec3 v, vec3 dv)
{
	float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
	float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
	vec3 d;
	d.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
	d.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
	d.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
	return d;
}

/* CR/auxiliary.h:139-164.  Returns 0 = culled, 1 = in, -1 = "prefiltered" trap. */
static int in_frustum(int idx, const float* orig_points, const float* viewmatrix, int prefiltered, vec3* p_view)
{
	vec3 p_orig = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
	*p_view = transformPoint4x3(p_orig, viewmatrix);
	if (p_view->z <= 0.2f) return prefiltered ? -1 : 0;
	return 1;
}

/* ---------------------------------------------------------------- CR/forward.cu */
/* CR/forward.cu:20-71 */
static vec3 computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                               const float* shs, uint8_t* clamped)
{
	vec3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
	vec3 dir = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
	float len = sqrtf(dot3(dir, dir));
	dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
	const float* sh = shs + (size_t)idx * max_coeffs * 3;
	float res[3];
	float x = dir.x, y = dir.y, z = dir.z;
	for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k) * 3 + ch]
		float r = SH_C0 * SH(0);
		if (deg > 0) {
			r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
			if (deg > 1) {
				float xx = x * x, yy = y * y, zz = z * z;
				float xy = x * y, yz = y * z, xz = x * z;
				r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
				    SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
				if (deg > 2) {
					r = r + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
					    SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
					    SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
					    SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
					    SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
				}
			}
		}
#undef SH
		r += 0.5f;
		clamped[3 * idx + ch] = (r < 0);
		res[ch] = r > 0.0f ? r : 0.0f; /* glm::max(result, 0.0f) */
	}
	vec3 out = {res[0], res[1], res[2]};
	return out;
}

/* shared by CR/forward.cu:74-113 and CR/backward.cu:162-194 */
static void cov2d_setup(vec3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, const float* cov3D,
                        const float* viewmatrix, vec3* t_out, float* xmul, float* ymul, mat3* W_out, mat3* T_out,
                        mat3* Vrk_out)
{
	vec3 t = transformPoint4x3(mean, viewmatrix);
	const float limx = 1.3f * tan_fovx;
	const float limy = 1.3f * tan_fovy;
	const float txtz = t.x / t.z;
	const float tytz = t.y / t.z;
	t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
	t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
	*xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f; /* CR/backward.cu:175 */
	*ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f; /* CR/backward.cu:176 */
	mat3 J = m3(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z,
	            -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
	mat3 W = m3(viewmatrix[0], viewmatrix[4], viewmatrix[8], viewmatrix[1], viewmatrix[5], viewmatrix[9],
	            viewmatrix[2], viewmatrix[6], viewmatrix[10]);
	*T_out = m3_mul(W, J);
	*Vrk_out = m3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
	*W_out = W;
	*t_out = t;
}

/* CR/forward.cu:74-113 */
static void computeCov2D(vec3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, const float* cov3D,
                         const float* viewmatrix, float* cov)
{
	vec3 t; float xm, ym; mat3 W, T, Vrk;
	cov2d_setup(mean, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &t, &xm, &ym, &W, &T, &Vrk);
	mat3 c = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
	c.c[0][0] += 0.3f;
	c.c[1][1] += 0.3f;
	cov[0] = c.c[0][0]; cov[1] = c.c[0][1]; cov[2] = c.c[1][1];
}

static mat3 quat_to_R(const float* rot)
{
	float r = rot[0], x = rot[1], y = rot[2], z = rot[3]; /* not normalised: CR/forward.cu:127 */
	return m3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y), 2.f * (x * y + r * z),
	          1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x), 2.f * (x * z - r * y), 2.f * (y * z + r * x),
	          1.f - 2.f * (x * x + y * y));
}

/* CR/forward.cu:118-152 */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
	mat3 S = m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
	S.c[0][0] = mod * scale[0];
	S.c[1][1] = mod * scale[1];
	S.c[2][2] = mod * scale[2];
	mat3 R = quat_to_R(rot);
	mat3 M = m3_mul(S, R);
	mat3 Sigma = m3_mul(m3_t(M), M);
	cov3D[0] = Sigma.c[0][0]; cov3D[1] = Sigma.c[0][1]; cov3D[2] = Sigma.c[0][2];
	cov3D[3] = Sigma.c[1][1]; cov3D[4] = Sigma.c[1][2]; cov3D[5] = Sigma.c[2][2];
}

/*
 * CR/forward.cu:155-256 (preprocessCUDA) and, with filter_only != 0, CR/forward.cu:260-335
 * (filter_preprocessCUDA: same path, writes radii only).  Returns 0, or -1 if a point is
 * culled although `prefiltered` is set (the reference printf+__trap()s, CR/auxiliary.h:156-160).
 * Per-Gaussian outputs of culled Gaussians are left untouched (the reference leaves them
 * uninitialised); callers zero them first.
 */
int bsro_preprocess(int P, int D, int M, const float* orig_points, const float* scales, float scale_modifier,
                    const float* rotations, const float* opacities, const float* shs, uint8_t* clamped,
                    const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                    const float* projmatrix, const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy,
                    int* radii, float* points_xy_image, float* depths, float* cov3Ds, float* rgb,
                    float* conic_opacity, uint32_t* tiles_touched, int prefiltered, int filter_only)
{
	/* CR/rasterizer_impl.cu:223-224,234 */
	const float focal_y = H / (2.0f * tan_fovy);
	const float focal_x = W / (2.0f * tan_fovx);
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	int trap = 0;
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++) {
		radii[idx] = 0;
		if (!filter_only) tiles_touched[idx] = 0;
		vec3 p_view;
		int inf = in_frustum(idx, orig_points, viewmatrix, prefiltered, &p_view);
		if (inf < 0) {
#pragma omp atomic write
			trap = 1;
			continue;
		}
		if (!inf) continue;
		vec3 p_orig = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
		float p_hom[4];
		transformPoint4x4(p_orig, projmatrix, p_hom);
		float p_w = 1.0f / (p_hom[3] + 0.0000001f);
		float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
		const float* cov3D;
		if (cov3D_precomp != NULL) {
			cov3D = cov3D_precomp + (size_t)idx * 6;
		} else {
			computeCov3D(scales + (size_t)idx * 3, scale_modifier, rotations + (size_t)idx * 4, cov3Ds + (size_t)idx * 6);
			cov3D = cov3Ds + (size_t)idx * 6;
		}
		float cov[3];
		computeCov2D(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, cov);
		float det = (cov[0] * cov[2] - cov[1] * cov[1]);
		if (det == 0.0f) continue;
		float det_inv = 1.f / det;
		float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
		float mid = 0.5f * (cov[0] + cov[2]);
		float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
		float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
		float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
		float pix[2] = {ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H)};
		int rmin[2], rmax[2];
		getRect(pix[0], pix[1], (int)my_radius, gx, gy, rmin, rmax);
		if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
		if (filter_only) { radii[idx] = (int)my_radius; continue; }
		if (colors_precomp == NULL) {
			vec3 c = computeColorFromSH(idx, D, M, orig_points, cam_pos, shs, clamped);
			rgb[idx * 3 + 0] = c.x; rgb[idx * 3 + 1] = c.y; rgb[idx * 3 + 2] = c.z;
		}
		depths[idx] = p_view.z;
		radii[idx] = (int)my_radius;
		points_xy_image[2 * idx] = pix[0];
		points_xy_image[2 * idx + 1] = pix[1];
		conic_opacity[4 * idx + 0] = conic[0];
		conic_opacity[4 * idx + 1] = conic[1];
		conic_opacity[4 * idx + 2] = conic[2];
		conic_opacity[4 * idx + 3] = opacities[idx];
		tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
	}
	return trap ? -1 : 0;
}

/* CR/rasterizer_impl.cu:54-66 (checkFrustum with prefiltered=false) */
void bsro_mark_visible(int P, const float* orig_points, const float* viewmatrix, const float* projmatrix,
                       uint8_t* present)
{
	(void)projmatrix;
	for (int idx = 0; idx < P; idx++) {
		vec3 pv;
		present[idx] = (uint8_t)(in_frustum(idx, orig_points, viewmatrix, 0, &pv) == 1);
	}
}

/* ---------------------------------------------------------------- CR/rasterizer_impl.cu binning */
/* cub::DeviceScan::InclusiveSum, CR/rasterizer_impl.cu:278; returns num_rendered (:282). */
uint32_t bsro_inclusive_sum(int P, const uint32_t* in, uint32_t* out)
{
	uint32_t s = 0;
	for (int i = 0; i < P; i++) { s += in[i]; out[i] = s; }
	return P > 0 ? out[P - 1] : 0;
}

/* CR/rasterizer_impl.cu:35-50 */
uint32_t bsro_get_higher_msb(uint32_t n)
{
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1) {
		step /= 2;
		if (n >> msb) msb += step; else msb -= step;
	}
	if (n >> msb) msb++;
	return msb;
}

/* CR/rasterizer_impl.cu:70-111 */
void bsro_duplicate_with_keys(int P, const float* points_xy, const float* depths, const uint32_t* offsets,
                              uint64_t* keys_unsorted, uint32_t* values_unsorted, const int* radii, int gx, int gy)
{
	for (int idx = 0; idx < P; idx++) {
		if (radii[idx] > 0) {
			uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
			int rmin[2], rmax[2];
			getRect(points_xy[2 * idx], points_xy[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
			for (int y = rmin[1]; y < rmax[1]; y++) {
				for (int x = rmin[0]; x < rmax[0]; x++) {
					uint64_t key = (uint64_t)(y * gx + x);
					key <<= 32;
					uint32_t dbits;
					memcpy(&dbits, &depths[idx], 4);
					key |= dbits;
					keys_unsorted[off] = key;
					values_unsorted[off] = (uint32_t)idx;
					off++;
				}
			}
		}
	}
}

/* cub::DeviceRadixSort::SortPairs(begin_bit=0, end_bit): stable LSD radix sort on the low
 * end_bit bits of the key.  CR/rasterizer_impl.cu:304-309. */
void bsro_sort_pairs(int n, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                     int end_bit)
{
	uint64_t* ka = (uint64_t*)malloc((size_t)(n > 0 ? n : 1) * 8);
	uint64_t* kb = (uint64_t*)malloc((size_t)(n > 0 ? n : 1) * 8);
	uint32_t* va = (uint32_t*)malloc((size_t)(n > 0 ? n : 1) * 4);
	uint32_t* vb = (uint32_t*)malloc((size_t)(n > 0 ? n : 1) * 4);
	memcpy(ka, keys_in, (size_t)n * 8);
	memcpy(va, vals_in, (size_t)n * 4);
	for (int shift = 0; shift < end_bit; shift += 8) {
		int bits = end_bit - shift < 8 ? end_bit - shift : 8;
		uint32_t mask = (1u << bits) - 1u;
		size_t count[257];
		memset(count, 0, sizeof(count));
		for (int i = 0; i < n; i++) count[((ka[i] >> shift) & mask) + 1]++;
		for (int b = 0; b < 256; b++) count[b + 1] += count[b];
		for (int i = 0; i < n; i++) {
			size_t d = count[(ka[i] >> shift) & mask]++;
			kb[d] = ka[i];
			vb[d] = va[i];
		}
		uint64_t* tk = ka; ka = kb; kb = tk;
		uint32_t* tv = va; va = vb; vb = tv;
	}
	memcpy(keys_out, ka, (size_t)n * 8);
	memcpy(vals_out, va, (size_t)n * 4);
	free(ka); free(kb); free(va); free(vb);
}

/* CR/rasterizer_impl.cu:116-138; ranges must be zeroed first (:311). ranges is uint2[tiles]. */
void bsro_identify_tile_ranges(int L, const uint64_t* keys, uint32_t* ranges)
{
	for (int idx = 0; idx < L; idx++) {
		uint32_t currtile = (uint32_t)(keys[idx] >> 32);
		if (idx == 0)
			ranges[2 * currtile] = 0;
		else {
			uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
			if (currtile != prevtile) {
				ranges[2 * prevtile + 1] = (uint32_t)idx;
				ranges[2 * currtile] = (uint32_t)idx;
			}
		}
		if (idx == L - 1) ranges[2 * currtile + 1] = (uint32_t)L;
	}
}

/* ---------------------------------------------------------------- CR/forward.cu:341-471 */
void bsro_render_forward(const uint32_t* ranges, const uint32_t* point_list, int W, int H,
                         const float* points_xy_image, const float* features, const float* depths,
                         const float* conic_opacity, float* final_T, uint32_t* n_contrib, const float* bg_color,
                         float* out_color, float* out_depth)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
	for (int tile = 0; tile < gx * gy; tile++) {
		const int tx = tile % gx, ty = tile / gx;
		const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
		for (int ly = 0; ly < BLOCK_Y; ly++)
			for (int lx = 0; lx < BLOCK_X; lx++) {
				const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
				if (!(px < W && py < H)) continue; /* inside, :354 */
				const uint32_t pix_id = (uint32_t)W * py + px;
				const float pixf[2] = {(float)px, (float)py};
				float T = 1.0f;
				uint32_t contributor = 0, last_contributor = 0;
				float C[NUM_CHANNELS] = {0};
				float Dd = 0;
				float acc = 0.000001f;
				for (uint32_t s = r0; s < r1; s++) {
					contributor++;
					const uint32_t id = point_list[s];
					const float dx = points_xy_image[2 * id] - pixf[0];
					const float dy = points_xy_image[2 * id + 1] - pixf[1];
					const float* con_o = conic_opacity + 4 * (size_t)id;
					float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
					if (power > 0.0f) continue;
					float alpha = fminf(0.99f, con_o[3] * bsro_expf(power));
					if (alpha < 1.0f / 255.0f) continue;
					float test_T = T * (1 - alpha);
					if (test_T < 0.0001f) break; /* done = true, :433-437 */
					/* `C[ch] += features[...] * alpha * T;` etc. (:439-446).  The reference is built with nvcc's defaults
					 * (RAST/setup.py passes no -fmad=false), under which a product that feeds an add is contracted:
					 * these three statements are x = f * alpha (rounded), then ONE fused multiply-add with T.  They are
					 * restated that way; every other expression of this file stays in source order without contraction
					 * (where nvcc's choice among several possible contractions is not determined by the source). */
					for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] = fmaf(features[id * NUM_CHANNELS + ch] * alpha, T, C[ch]);
					Dd = fmaf(depths[id] * alpha, T, Dd);
					acc = fmaf(alpha, T, acc);
					T = test_T;
					last_contributor = contributor;
				}
				final_T[pix_id] = T;
				n_contrib[pix_id] = last_contributor;
				for (int ch = 0; ch < NUM_CHANNELS; ch++)
					out_color[(size_t)ch * H * W + pix_id] = fmaf(T, bg_color[ch], C[ch]); /* :462, contracted likewise */
				out_depth[pix_id] = (acc > 0.5f) ? Dd / acc : 0; /* :464-468 */
			}
	}
}

/* ---------------------------------------------------------------- CR/backward.cu:399-586 */
/*
 * The reference accumulates 9 floats per (pixel, Gaussian) with float atomicAdd in an
 * unspecified order (:537,574-583).  Here each list slot gets a binary64 partial sum over its
 * tile's pixels (pixel order), slots are then added per Gaussian in list order, and the result
 * is rounded to binary32 once.  abs_sums (optional, [P*9]) receives sum(|term|) so tests can
 * bound the legitimate reordering error.  dL_dmean2D is [P,3], dL_dconic2D [P,4] (x,y,_,w),
 * dL_dopacity [P], dL_dcolors [P,3]; all must be zeroed by the caller (RAST/rasterize_points.cu:154-158).
 * dL_depths is accepted and ignored, as in the reference (:457-463,539-554 are commented out).
 */
/* Sum mode of bsro_render_backward.  0 (default): binary64 partial sums, rounded once -- the order-free
 * definition the goldens and the parity tests use.  1: every add is a binary32 add, slot by slot in pixel order
 * and then Gaussian by Gaussian in list order = what the reference's float atomicAdds (:537,574-583) produce when
 * they happen to execute in that order; ONE legal outcome of the reference, used only to measure how far two legal
 * outcomes of the reference lie apart (tools/parity_report.py, tests: the floor of any elementwise comparison). */
static int g_sum_f32 = 0;
void bsro_set_sum_mode(int f32) { g_sum_f32 = f32; }
#define ACC(dst, v) do { if (g_sum_f32) (dst) = (double)((float)(dst) + (float)(v)); else (dst) += (v); } while (0)

/* Measurement only (docs/EXPERIMENTS.md, round 6: "could a long tile's backward be cut into segments?"): with
 * seg > 0 every pixel's back-to-front recurrences are RESTARTED wherever the list position is a multiple of seg, from
 * what a forward pass could have checkpointed there -- T before the entry and the colour accumulated before it -- plus
 * the pixel's final colour: T <- T_ck, accum_rec <- (C_final - C_ck) / T_ck (a subtraction instead of the reference's
 * recurrence, :527-536).  Not the reference's arithmetic; used to measure how far such a walk would land from it. */
static int g_bwd_seg = 0;
void bsro_set_backward_segment(int seg) { g_bwd_seg = seg; }
#define BSRO_MAX_CK 4096

/* -DBSRO_CONTRACT_RENDER_BWD (make fma -> libbsr_oracle_fma.so, measurement only): THIS function alone is compiled with
 * floating-point contraction, i.e. the compiler may fuse a * b + c into one FMA wherever it likes -- what nvcc does to
 * the reference by default (RAST/setup.py passes no -fmad=false).  Both builds are legal evaluations of backward.cu:
 * 496-586; the distance between their results (tools/parity_report.py: "reference_contracted_vs_source_order") is the
 * floor for comparing ANY implementation's per-pair terms with this file's, next to the summation-order floor above. */
#ifdef BSRO_CONTRACT_RENDER_BWD
#define BSRO_RB_ATTR __attribute__((optimize("fp-contract=fast")))
#else
#define BSRO_RB_ATTR
#endif
BSRO_RB_ATTR
void bsro_render_backward(int P, int R, const uint32_t* ranges, const uint32_t* point_list, int W, int H,
                          const float* bg_color, const float* points_xy_image, const float* conic_opacity,
                          const float* colors, const float* final_Ts, const uint32_t* n_contrib,
                          const float* dL_dpixels, const float* dL_depths, float* dL_dmean2D, float* dL_dconic2D,
                          float* dL_dopacity, float* dL_dcolors, float* abs_sums)
{
	(void)dL_depths;
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	double* slab = (double*)calloc((size_t)(R > 0 ? R : 1) * 9, sizeof(double));
	double* aslab = abs_sums ? (double*)calloc((size_t)(R > 0 ? R : 1) * 9, sizeof(double)) : NULL;
	const float ddelx_dx = 0.5 * W; /* :473 */
	const float ddely_dy = 0.5 * H; /* :474 */
#pragma omp parallel for schedule(dynamic, 4)
	for (int tile = 0; tile < gx * gy; tile++) {
		const int tx = tile % gx, ty = tile / gx;
		const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
		for (int ly = 0; ly < BLOCK_Y; ly++)
			for (int lx = 0; lx < BLOCK_X; lx++) {
				const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
				if (!(px < W && py < H)) continue;
				const uint32_t pix_id = (uint32_t)W * py + px;
				const float pixf[2] = {(float)px, (float)py};
				const float T_final = final_Ts[pix_id];
				float T = T_final;
				uint32_t contributor = r1 - r0;
				const int last_contributor = (int)n_contrib[pix_id];
				float accum_rec[NUM_CHANNELS] = {0};
				float dL_dpixel[NUM_CHANNELS];
				for (int i = 0; i < NUM_CHANNELS; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
				float last_alpha = 0;
				float last_color[NUM_CHANNELS] = {0};
				/* (measurement mode) forward checkpoints of this pixel at the list positions that are multiples of seg */
				float ckT[BSRO_MAX_CK], ckC[BSRO_MAX_CK][NUM_CHANNELS], Cfin[NUM_CHANNELS] = {0};
				const int seg = g_bwd_seg;
				if (seg > 0) {
					float Tf = 1.0f, Cf[NUM_CHANNELS] = {0};
					for (uint32_t q = 0; q < (uint32_t)last_contributor; q++) {
						if (q % (uint32_t)seg == 0 && q / (uint32_t)seg < BSRO_MAX_CK) {
							ckT[q / seg] = Tf;
							for (int ch = 0; ch < NUM_CHANNELS; ch++) ckC[q / seg][ch] = Cf[ch];
						}
						const uint32_t idf = point_list[r0 + q];
						const float dxf = points_xy_image[2 * idf] - pixf[0], dyf = points_xy_image[2 * idf + 1] - pixf[1];
						const float* cf = conic_opacity + 4 * (size_t)idf;
						const float pw = -0.5f * (cf[0] * dxf * dxf + cf[2] * dyf * dyf) - cf[1] * dxf * dyf;
						if (pw > 0.0f) continue;
						const float af = fminf(0.99f, cf[3] * bsro_expf(pw));
						if (af < 1.0f / 255.0f) continue;
						for (int ch = 0; ch < NUM_CHANNELS; ch++) Cf[ch] = fmaf(colors[idf * NUM_CHANNELS + ch] * af, Tf, Cf[ch]);
						Tf = Tf * (1 - af);
					}
					for (int ch = 0; ch < NUM_CHANNELS; ch++) Cfin[ch] = Cf[ch];
				}
				for (uint32_t k = 0; k < r1 - r0; k++) {
					const uint32_t slot = r1 - k - 1; /* back to front, :485 */
					contributor--;
					if (contributor >= (uint32_t)last_contributor) continue;
					if (seg > 0 && (contributor + 1) % (uint32_t)seg == 0 && contributor + 1 < (uint32_t)last_contributor &&
					    (contributor + 1) / (uint32_t)seg < BSRO_MAX_CK) {
						/* first entry of a segment (walking backwards): restart from the checkpoint behind it */
						const uint32_t b = (contributor + 1) / (uint32_t)seg;
						T = ckT[b];
						for (int ch = 0; ch < NUM_CHANNELS; ch++) accum_rec[ch] = (Cfin[ch] - ckC[b][ch]) / ckT[b];
						last_alpha = 0;
					}
					const uint32_t id = point_list[slot];
					const float dx = points_xy_image[2 * id] - pixf[0];
					const float dy = points_xy_image[2 * id + 1] - pixf[1];
					const float* con_o = conic_opacity + 4 * (size_t)id;
					const float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
					if (power > 0.0f) continue;
					const float G = bsro_expf(power);
					const float alpha = fminf(0.99f, con_o[3] * G);
					if (alpha < 1.0f / 255.0f) continue;
					T = T / (1.f - alpha);
					const float dchannel_dcolor = alpha * T;
					float dL_dalpha = 0.0f;
					double* sl = slab + (size_t)slot * 9;
					double* asl = aslab ? aslab + (size_t)slot * 9 : NULL;
					for (int ch = 0; ch < NUM_CHANNELS; ch++) {
						const float c = colors[id * NUM_CHANNELS + ch];
						accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
						last_color[ch] = c;
						const float dL_dchannel = dL_dpixel[ch];
						dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
						const float v = dchannel_dcolor * dL_dchannel;
						ACC(sl[6 + ch], v);
						if (asl) asl[6 + ch] += fabs((double)v);
					}
					dL_dalpha *= T;
					last_alpha = alpha;
					float bg_dot_dpixel = 0;
					for (int i = 0; i < NUM_CHANNELS; i++) bg_dot_dpixel += bg_color[i] * dL_dpixel[i];
					dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
					const float dL_dG = con_o[3] * dL_dalpha;
					const float gdx = G * dx;
					const float gdy = G * dy;
					const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
					const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
					const float v0 = dL_dG * dG_ddelx * ddelx_dx;
					const float v1 = dL_dG * dG_ddely * ddely_dy;
					const float v2 = -0.5f * gdx * dx * dL_dG;
					const float v3 = -0.5f * gdx * dy * dL_dG;
					const float v4 = -0.5f * gdy * dy * dL_dG;
					const float v5 = G * dL_dalpha;
					ACC(sl[0], v0); ACC(sl[1], v1); ACC(sl[2], v2); ACC(sl[3], v3); ACC(sl[4], v4); ACC(sl[5], v5);
					if (asl) {
						asl[0] += fabs((double)v0); asl[1] += fabs((double)v1); asl[2] += fabs((double)v2);
						asl[3] += fabs((double)v3); asl[4] += fabs((double)v4); asl[5] += fabs((double)v5);
					}
				}
			}
	}
	double* acc = (double*)calloc((size_t)(P > 0 ? P : 1) * 9, sizeof(double));
	double* aacc = abs_sums ? (double*)calloc((size_t)(P > 0 ? P : 1) * 9, sizeof(double)) : NULL;
	for (int s = 0; s < R; s++) {
		const uint32_t id = point_list[s];
		for (int k = 0; k < 9; k++) ACC(acc[(size_t)id * 9 + k], slab[(size_t)s * 9 + k]);
		if (aacc)
			for (int k = 0; k < 9; k++) aacc[(size_t)id * 9 + k] += aslab[(size_t)s * 9 + k];
	}
	for (int i = 0; i < P; i++) {
		const double* a = acc + (size_t)i * 9;
		dL_dmean2D[3 * i + 0] += (float)a[0];
		dL_dmean2D[3 * i + 1] += (float)a[1];
		dL_dconic2D[4 * i + 0] += (float)a[2];
		dL_dconic2D[4 * i + 1] += (float)a[3];
		dL_dconic2D[4 * i + 3] += (float)a[4];
		dL_dopacity[i] += (float)a[5];
		dL_dcolors[3 * i + 0] += (float)a[6];
		dL_dcolors[3 * i + 1] += (float)a[7];
		dL_dcolors[3 * i + 2] += (float)a[8];
		if (aacc)
			for (int k = 0; k < 9; k++) abs_sums[(size_t)i * 9 + k] = (float)aacc[(size_t)i * 9 + k];
	}
	free(slab); free(acc);
	if (aslab) free(aslab);
	if (aacc) free(aacc);
}

/* ---------------------------------------------------------------- EXTENSION (not in the reference)
 * Opt-in gradient of the depth target.  The reference forward normalises
 * depth = D / acc (acc > 0.5, else 0; CR/forward.cu:459-468) but its backward drops dL_ddepth
 * entirely (CR/backward.cu:457-463,539-554 are commented out, and written for another depth
 * definition).  This is the TRUE derivative of the shipped forward, restated in the same walk as
 * bsro_render_backward; its pin is the float64 autograd of oracle/torch_splat.py, not the reference.
 *   w_i = alpha_i T_i,  D = sum z_i w_i,  A = 1e-6 + sum w_i,  depth = D / A  (A > 0.5)
 *   gz = g / A,  g1 = -g depth / A,  d_i = gz z_i + g1          (g = dL/d depth of the pixel)
 *   dL/dz_i     = gz w_i
 *   dL/dalpha_i = T_i (d_i - Rd_i),  Rd_{i-1} = alpha_i d_i + (1 - alpha_i) Rd_i   (back to front)
 * and dL/dalpha_i feeds mean2D / conic / opacity exactly like the colour term (no background term).
 * The gate is the forward's own decision (out_depth != 0: view z > 0.2 makes D > 0 whenever
 * A > 0.5) and A is rebuilt as 1e-6 + (1 - T_final).  ADDS to dL_dmean2D [P,3], dL_dconic2D [P,4],
 * dL_dopacity [P] (binary64 partial sums, rounded once) and writes dL_dz [P]. */
void bsro_render_backward_depth(int P, int R, const uint32_t* ranges, const uint32_t* point_list, int W, int H,
                                const float* points_xy_image, const float* conic_opacity, const float* depths,
                                const float* final_Ts, const uint32_t* n_contrib, const float* out_depth,
                                const float* dL_depths, float* dL_dmean2D, float* dL_dconic2D, float* dL_dopacity,
                                float* dL_dz)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	double* slab = (double*)calloc((size_t)(R > 0 ? R : 1) * 7, sizeof(double));
	const float ddelx_dx = 0.5 * W;
	const float ddely_dy = 0.5 * H;
#pragma omp parallel for schedule(dynamic, 4)
	for (int tile = 0; tile < gx * gy; tile++) {
		const int tx = tile % gx, ty = tile / gx;
		const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
		for (int ly = 0; ly < BLOCK_Y; ly++)
			for (int lx = 0; lx < BLOCK_X; lx++) {
				const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
				if (!(px < W && py < H)) continue;
				const uint32_t pix_id = (uint32_t)W * py + px;
				const float depth_px = out_depth[pix_id];
				if (!(depth_px != 0.0f)) continue; /* acc <= 0.5: the forward wrote a constant 0 */
				const float pixf[2] = {(float)px, (float)py};
				const float T_final = final_Ts[pix_id];
				const float A = 1e-6f + (1.0f - T_final);
				const float gz = dL_depths[pix_id] / A;
				const float g1 = -gz * depth_px;
				float T = T_final;
				uint32_t contributor = r1 - r0;
				const int last_contributor = (int)n_contrib[pix_id];
				float Rd = 0.0f;
				for (uint32_t k = 0; k < r1 - r0; k++) {
					const uint32_t slot = r1 - k - 1;
					contributor--;
					if (contributor >= (uint32_t)last_contributor) continue;
					const uint32_t id = point_list[slot];
					const float dx = points_xy_image[2 * id] - pixf[0];
					const float dy = points_xy_image[2 * id + 1] - pixf[1];
					const float* con_o = conic_opacity + 4 * (size_t)id;
					const float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
					if (power > 0.0f) continue;
					const float G = bsro_expf(power);
					const float alpha = fminf(0.99f, con_o[3] * G);
					if (alpha < 1.0f / 255.0f) continue;
					T = T / (1.f - alpha);
					const float d_i = gz * depths[id] + g1;
					const float e = d_i - Rd;
					Rd = Rd + alpha * e;
					const float dL_dalpha = T * e;
					const float dL_dG = con_o[3] * dL_dalpha;
					const float gdx = G * dx;
					const float gdy = G * dy;
					const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
					const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
					double* sl = slab + (size_t)slot * 7;
					sl[0] += dL_dG * dG_ddelx * ddelx_dx;
					sl[1] += dL_dG * dG_ddely * ddely_dy;
					sl[2] += -0.5f * gdx * dx * dL_dG;
					sl[3] += -0.5f * gdx * dy * dL_dG;
					sl[4] += -0.5f * gdy * dy * dL_dG;
					sl[5] += G * dL_dalpha;
					sl[6] += gz * (alpha * T);
				}
			}
	}
	double* acc = (double*)calloc((size_t)(P > 0 ? P : 1) * 7, sizeof(double));
	for (int s = 0; s < R; s++) {
		const uint32_t id = point_list[s];
		for (int k = 0; k < 7; k++) acc[(size_t)id * 7 + k] += slab[(size_t)s * 7 + k];
	}
	for (int i = 0; i < P; i++) {
		const double* a = acc + (size_t)i * 7;
		dL_dmean2D[3 * i + 0] += (float)a[0];
		dL_dmean2D[3 * i + 1] += (float)a[1];
		dL_dconic2D[4 * i + 0] += (float)a[2];
		dL_dconic2D[4 * i + 1] += (float)a[3];
		dL_dconic2D[4 * i + 3] += (float)a[4];
		dL_dopacity[i] += (float)a[5];
		dL_dz[i] = (float)a[6];
	}
	free(slab);
	free(acc);
}

/* ---------------------------------------------------------------- CR/backward.cu:144-274 */
void bsro_backward_cov2d(int P, const float* means, const int* radii, const float* cov3Ds, float h_x, float h_y,
                         float tan_fovx, float tan_fovy, const float* view_matrix, const float* dL_dconics,
                         float* dL_dmeans, float* dL_dcov)
{
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++) {
		if (!(radii[idx] > 0)) continue;
		const float* cov3D = cov3Ds + 6 * (size_t)idx;
		vec3 mean = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
		float dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
		vec3 t; float x_grad_mul, y_grad_mul; mat3 W, T, Vrk;
		cov2d_setup(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view_matrix, &t, &x_grad_mul, &y_grad_mul, &W, &T, &Vrk);
		mat3 cov2D = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
		float a = cov2D.c[0][0] += 0.3f;
		float b = cov2D.c[0][1];
		float c = cov2D.c[1][1] += 0.3f;
		float denom = a * c - b * b;
		float dL_da = 0, dL_db = 0, dL_dc = 0;
		float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define Tm(i, j) T.c[i][j]
#define Vm(i, j) Vrk.c[i][j]
#define Wm(i, j) W.c[i][j]
		if (denom2inv != 0) {
			dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
			dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
			dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
			dL_dcov[6 * idx + 0] = (Tm(0,0) * Tm(0,0) * dL_da + Tm(0,0) * Tm(1,0) * dL_db + Tm(1,0) * Tm(1,0) * dL_dc);
			dL_dcov[6 * idx + 3] = (Tm(0,1) * Tm(0,1) * dL_da + Tm(0,1) * Tm(1,1) * dL_db + Tm(1,1) * Tm(1,1) * dL_dc);
			dL_dcov[6 * idx + 5] = (Tm(0,2) * Tm(0,2) * dL_da + Tm(0,2) * Tm(1,2) * dL_db + Tm(1,2) * Tm(1,2) * dL_dc);
			dL_dcov[6 * idx + 1] = 2 * Tm(0,0) * Tm(0,1) * dL_da + (Tm(0,0) * Tm(1,1) + Tm(0,1) * Tm(1,0)) * dL_db + 2 * Tm(1,0) * Tm(1,1) * dL_dc;
			dL_dcov[6 * idx + 2] = 2 * Tm(0,0) * Tm(0,2) * dL_da + (Tm(0,0) * Tm(1,2) + Tm(0,2) * Tm(1,0)) * dL_db + 2 * Tm(1,0) * Tm(1,2) * dL_dc;
			dL_dcov[6 * idx + 4] = 2 * Tm(0,2) * Tm(0,1) * dL_da + (Tm(0,1) * Tm(1,2) + Tm(0,2) * Tm(1,1)) * dL_db + 2 * Tm(1,1) * Tm(1,2) * dL_dc;
		} else {
			for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
		}
		float dL_dT00 = 2 * (Tm(0,0) * Vm(0,0) + Tm(0,1) * Vm(0,1) + Tm(0,2) * Vm(0,2)) * dL_da +
		                (Tm(1,0) * Vm(0,0) + Tm(1,1) * Vm(0,1) + Tm(1,2) * Vm(0,2)) * dL_db;
		float dL_dT01 = 2 * (Tm(0,0) * Vm(1,0) + Tm(0,1) * Vm(1,1) + Tm(0,2) * Vm(1,2)) * dL_da +
		                (Tm(1,0) * Vm(1,0) + Tm(1,1) * Vm(1,1) + Tm(1,2) * Vm(1,2)) * dL_db;
		float dL_dT02 = 2 * (Tm(0,0) * Vm(2,0) + Tm(0,1) * Vm(2,1) + Tm(0,2) * Vm(2,2)) * dL_da +
		                (Tm(1,0) * Vm(2,0) + Tm(1,1) * Vm(2,1) + Tm(1,2) * Vm(2,2)) * dL_db;
		float dL_dT10 = 2 * (Tm(1,0) * Vm(0,0) + Tm(1,1) * Vm(0,1) + Tm(1,2) * Vm(0,2)) * dL_dc +
		                (Tm(0,0) * Vm(0,0) + Tm(0,1) * Vm(0,1) + Tm(0,2) * Vm(0,2)) * dL_db;
		float dL_dT11 = 2 * (Tm(1,0) * Vm(1,0) + Tm(1,1) * Vm(1,1) + Tm(1,2) * Vm(1,2)) * dL_dc +
		                (Tm(0,0) * Vm(1,0) + Tm(0,1) * Vm(1,1) + Tm(0,2) * Vm(1,2)) * dL_db;
		float dL_dT12 = 2 * (Tm(1,0) * Vm(2,0) + Tm(1,1) * Vm(2,1) + Tm(1,2) * Vm(2,2)) * dL_dc +
		                (Tm(0,0) * Vm(2,0) + Tm(0,1) * Vm(2,1) + Tm(0,2) * Vm(2,2)) * dL_db;
		float dL_dJ00 = Wm(0,0) * dL_dT00 + Wm(0,1) * dL_dT01 + Wm(0,2) * dL_dT02;
		float dL_dJ02 = Wm(2,0) * dL_dT00 + Wm(2,1) * dL_dT01 + Wm(2,2) * dL_dT02;
		float dL_dJ11 = Wm(1,0) * dL_dT10 + Wm(1,1) * dL_dT11 + Wm(1,2) * dL_dT12;
		float dL_dJ12 = Wm(2,0) * dL_dT10 + Wm(2,1) * dL_dT11 + Wm(2,2) * dL_dT12;
#undef Tm
#undef Vm
#undef Wm
		float tz = 1.f / t.z;
		float tz2 = tz * tz;
		float tz3 = tz2 * tz;
		float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
		float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
		float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
		               (2 * h_y * t.y) * tz3 * dL_dJ12;
		vec3 d = {dL_dtx, dL_dty, dL_dtz};
		vec3 dL_dmean = transformVec4x3Transpose(d, view_matrix);
		dL_dmeans[3 * idx] = dL_dmean.x; /* assignment, :273 */
		dL_dmeans[3 * idx + 1] = dL_dmean.y;
		dL_dmeans[3 * idx + 2] = dL_dmean.z;
	}
}

/* CR/backward.cu:20-139 */
static void computeColorFromSH_bw(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                                  const float* shs, const uint8_t* clamped, const float* dL_dcolor, float* dL_dmeans,
                                  float* dL_dshs)
{
	vec3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
	vec3 dir_orig = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
	float len = sqrtf(dot3(dir_orig, dir_orig));
	vec3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
	const float* sh = shs + (size_t)idx * max_coeffs * 3;
	float dL_dRGB[3] = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
	dL_dRGB[0] *= clamped[3 * idx + 0] ? 0 : 1;
	dL_dRGB[1] *= clamped[3 * idx + 1] ? 0 : 1;
	dL_dRGB[2] *= clamped[3 * idx + 2] ? 0 : 1;
	float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
	float x = dir.x, y = dir.y, z = dir.z;
	float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SH(k) sh[(k) * 3 + ch]
#define DSH(k, v) do { float _v = (v); for (int ch = 0; ch < 3; ch++) dL_dsh[(k) * 3 + ch] = _v * dL_dRGB[ch]; } while (0)
	DSH(0, SH_C0);
	if (deg > 0) {
		DSH(1, -SH_C1 * y);
		DSH(2, SH_C1 * z);
		DSH(3, -SH_C1 * x);
		for (int ch = 0; ch < 3; ch++) {
			dRGBdx[ch] = -SH_C1 * SH(3);
			dRGBdy[ch] = -SH_C1 * SH(1);
			dRGBdz[ch] = SH_C1 * SH(2);
		}
		if (deg > 1) {
			float xx = x * x, yy = y * y, zz = z * z;
			float xy = x * y, yz = y * z, xz = x * z;
			DSH(4, SH_C2[0] * xy);
			DSH(5, SH_C2[1] * yz);
			DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
			DSH(7, SH_C2[3] * xz);
			DSH(8, SH_C2[4] * (xx - yy));
			for (int ch = 0; ch < 3; ch++) {
				dRGBdx[ch] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
				dRGBdy[ch] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
				dRGBdz[ch] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
			}
			if (deg > 2) {
				DSH(9, SH_C3[0] * y * (3.f * xx - yy));
				DSH(10, SH_C3[1] * xy * z);
				DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
				DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
				DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy));
				DSH(14, SH_C3[5] * z * (xx - yy));
				DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
				for (int ch = 0; ch < 3; ch++) {
					dRGBdx[ch] += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz +
					               SH_C3[2] * SH(11) * -2.f * xy + SH_C3[3] * SH(12) * -3.f * 2.f * xz +
					               SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14) * 2.f * xz +
					               SH_C3[6] * SH(15) * 3.f * (xx - yy));
					dRGBdy[ch] += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
					               SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
					               SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
					               SH_C3[6] * SH(15) * -3.f * 2.f * xy);
					dRGBdz[ch] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
					               SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
					               SH_C3[5] * SH(14) * (xx - yy));
				}
			}
		}
	}
#undef SH
#undef DSH
	vec3 dL_ddir = {col_dot(dRGBdx, dL_dRGB), col_dot(dRGBdy, dL_dRGB), col_dot(dRGBdz, dL_dRGB)};
	vec3 dL_dmean = dnormvdv(dir_orig, dL_ddir);
	dL_dmeans[3 * idx] += dL_dmean.x;
	dL_dmeans[3 * idx + 1] += dL_dmean.y;
	dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

/* CR/backward.cu:278-341 */
static void computeCov3D_bw(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                            float* dL_dscales, float* dL_drots)
{
	float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
	mat3 R = quat_to_R(rot);
	mat3 S = m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
	float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
	S.c[0][0] = s[0]; S.c[1][1] = s[1]; S.c[2][2] = s[2];
	mat3 M = m3_mul(S, R);
	const float* d = dL_dcov3Ds + 6 * (size_t)idx;
	mat3 dL_dSigma = m3(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
	/* dL_dM = 2.0f * M * dL_dSigma: (2.0f * M) first (scalar * mat), then mat * mat */
	mat3 M2;
	for (int c = 0; c < 3; c++)
		for (int rr = 0; rr < 3; rr++) M2.c[c][rr] = 2.0f * M.c[c][rr];
	mat3 dL_dM = m3_mul(M2, dL_dSigma);
	mat3 Rt = m3_t(R);
	mat3 dL_dMt = m3_t(dL_dM);
	float* dL_dscale = dL_dscales + 3 * (size_t)idx;
	dL_dscale[0] = col_dot(Rt.c[0], dL_dMt.c[0]);
	dL_dscale[1] = col_dot(Rt.c[1], dL_dMt.c[1]);
	dL_dscale[2] = col_dot(Rt.c[2], dL_dMt.c[2]);
	for (int k = 0; k < 3; k++) {
		dL_dMt.c[0][k] *= s[0];
		dL_dMt.c[1][k] *= s[1];
		dL_dMt.c[2][k] *= s[2];
	}
#define G(i, j) dL_dMt.c[i][j]
	float qx = 2 * z * (G(0,1) - G(1,0)) + 2 * y * (G(2,0) - G(0,2)) + 2 * x * (G(1,2) - G(2,1));
	float qy = 2 * y * (G(1,0) + G(0,1)) + 2 * z * (G(2,0) + G(0,2)) + 2 * r * (G(1,2) - G(2,1)) - 4 * x * (G(2,2) + G(1,1));
	float qz = 2 * x * (G(1,0) + G(0,1)) + 2 * r * (G(2,0) - G(0,2)) + 2 * z * (G(1,2) + G(2,1)) - 4 * y * (G(2,2) + G(0,0));
	float qw = 2 * r * (G(0,1) - G(1,0)) + 2 * x * (G(2,0) + G(0,2)) + 2 * y * (G(1,2) + G(2,1)) - 4 * z * (G(1,1) + G(0,0));
#undef G
	float* dL_drot = dL_drots + 4 * (size_t)idx;
	dL_drot[0] = qx; dL_drot[1] = qy; dL_drot[2] = qz; dL_drot[3] = qw;
}

/* CR/backward.cu:346-396 */
void bsro_backward_preprocess(int P, int D, int M, const float* means, const int* radii, const float* shs,
                              const uint8_t* clamped, const float* scales, const float* rotations,
                              float scale_modifier, const float* proj, const float* campos, const float* dL_dmean2D,
                              float* dL_dmeans, const float* dL_dcolor, const float* dL_dcov3D, float* dL_dsh,
                              float* dL_dscale, float* dL_drot)
{
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++) {
		if (!(radii[idx] > 0)) continue;
		vec3 m = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
		float m_hom[4];
		transformPoint4x4(m, proj, m_hom);
		float m_w = 1.0f / (m_hom[3] + 0.0000001f);
		float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
		float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
		const float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
		float dx = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
		float dy = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
		float dz = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
		dL_dmeans[3 * idx] += dx;
		dL_dmeans[3 * idx + 1] += dy;
		dL_dmeans[3 * idx + 2] += dz;
		if (shs) computeColorFromSH_bw(idx, D, M, means, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
		if (scales)
			computeCov3D_bw(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov3D, dL_dscale, dL_drot);
	}
}
