// Micro-benchmark: sustained VALU issue rate per SIMD on gfx950 for the instruction kinds the tile
// walks are made of (plain FMA, select on an SGPR mask, compare to SGPR, DPP add, packed FMA,
// rcp, ldexp, readfirstlane).  Build & run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rates tools/microbench/valu_rates.hip && /tmp/valu_rates
// Prints cycles per wave-instruction per SIMD at 8 waves/SIMD (2048 workgroups x 256 lanes), assuming
// the clock measured by s_memtime over the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters, float seed)
{
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const float b = 1.0001f, c = 0.5f;
	unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)(seed > 100.f);
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; i++) {
		if (KIND == 0) {
			asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
			                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
		} else if (KIND == 1) {
			asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
			                  "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(mask));
		} else if (KIND == 2) {
			asm volatile(REP8("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
			                  "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
			                  "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
			                  "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
		} else if (KIND == 3) {   // compare into an SGPR pair (VOP3 form), as the predicates of the walks do
			unsigned long long m0, m1;
			asm volatile(REP8("v_cmp_lt_f32_e64 %8, %0, %10\n v_cmp_lt_f32_e64 %9, %1, %10\n v_cmp_lt_f32_e64 %8, %2, %10\n v_cmp_lt_f32_e64 %9, %3, %10\n"
			                  "v_cmp_lt_f32_e64 %8, %4, %10\n v_cmp_lt_f32_e64 %9, %5, %10\n v_cmp_lt_f32_e64 %8, %6, %10\n v_cmp_lt_f32_e64 %9, %7, %10\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(m0), "=&s"(m1) : "v"(b));
			mask ^= m0 ^ m1;
		} else if (KIND == 4) {   // packed: two fp32 FMAs per lane per instruction
			typedef float f2 __attribute__((ext_vector_type(2)));
			f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
			asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
			                  "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
			             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
			a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
		} else if (KIND == 5) {
			asm volatile(REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
		} else if (KIND == 6) {
			int e = 1;
			asm volatile(REP8("v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %8\n v_ldexp_f32 %2, %2, %8\n v_ldexp_f32 %3, %3, %8\n v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n v_ldexp_f32 %6, %6, %8\n v_ldexp_f32 %7, %7, %8\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));
		} else if (KIND == 7) {   // plain VOP2 mul (for reference against the VOP3 fma)
			asm volatile(REP8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
		} else if (KIND == 8) {   // VALU and SALU interleaved 1:1, as in the compiled walks
			unsigned s0 = (unsigned)mask, s1 = 3;
			asm volatile(REP8("v_mul_f32 %0, %0, %9\n s_add_u32 %8, %8, %10\n v_mul_f32 %1, %1, %9\n s_add_u32 %8, %8, %10\n v_mul_f32 %2, %2, %9\n s_add_u32 %8, %8, %10\n v_mul_f32 %3, %3, %9\n s_add_u32 %8, %8, %10\n"
			                  "v_mul_f32 %4, %4, %9\n s_add_u32 %8, %8, %10\n v_mul_f32 %5, %5, %9\n s_add_u32 %8, %8, %10\n v_mul_f32 %6, %6, %9\n s_add_u32 %8, %8, %10\n v_mul_f32 %7, %7, %9\n s_add_u32 %8, %8, %10\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0) : "v"(b), "s"(s1) : "scc");
			mask ^= s0;
		} else if (KIND >= 9 && KIND <= 12) {   // v_fma_f32 under a partial EXEC mask: does a half (quarter) with no live lane cost a pass?
			const unsigned long long em = KIND == 9 ? 0x00000000ffffffffull : KIND == 10 ? 0x0000ffff0000ffffull
			                              : KIND == 11 ? 0x000000000000ffffull : 0x0000000100000001ull;
			unsigned long long save;
			asm volatile("s_mov_b64 %8, exec\n s_mov_b64 exec, %11\n"
			             REP8("v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n v_fma_f32 %3, %3, %9, %10\n"
			                  "v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n v_fma_f32 %6, %6, %9, %10\n v_fma_f32 %7, %7, %9, %10\n")
			             "s_mov_b64 exec, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(save) : "v"(b), "v"(c), "s"(em));
		} else if (KIND == 13) {   // v_exp_f32
			asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
		} else if (KIND == 14) {   // v_fma_f32 with one SGPR operand (the scalar-record walk's form)
			asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
			                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
		} else if (KIND == 15) {   // compare into VCC (VOPC e32) + select on VCC (VOP2 e32)
			asm volatile(REP8("v_cmp_lt_f32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %1, %8, vcc\n v_cmp_lt_f32_e32 vcc, %2, %8\n v_cndmask_b32_e32 %2, %3, %8, vcc\n"
			                  "v_cmp_lt_f32_e32 vcc, %4, %8\n v_cndmask_b32_e32 %4, %5, %8, vcc\n v_cmp_lt_f32_e32 vcc, %6, %8\n v_cndmask_b32_e32 %6, %7, %8, vcc\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
		} else if (KIND == 16) {   // v_readlane_b32 with an SGPR lane select
			unsigned s0 = 0, s1 = 5;
			asm volatile(REP8("v_readlane_b32 %8, %0, %9\n v_readlane_b32 %8, %1, %9\n v_readlane_b32 %8, %2, %9\n v_readlane_b32 %8, %3, %9\n"
			                  "v_readlane_b32 %8, %4, %9\n v_readlane_b32 %8, %5, %9\n v_readlane_b32 %8, %6, %9\n v_readlane_b32 %8, %7, %9\n")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0) : "s"(s1));
			mask ^= s0;
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(mask & 1);
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, int valu_per_iter)
{
	const int blocks = 256 * waves_per_simd;   // 4 waves per workgroup = 1 per SIMD of a CU
	const int iters = 2000;
	float* out; unsigned long long* cyc;
	hipMalloc(&out, (size_t)blocks * 256 * 4);
	hipMalloc(&cyc, (size_t)blocks * 8);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10, 1.0f);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> h(blocks);
	hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
	double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
	// s_memtime / readcyclecounter ticks at 100 MHz on gfx9: derive cycles from wall time at an assumed 2.4 GHz too
	const double instr_per_simd = (double)iters * valu_per_iter * waves_per_simd;
	printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD = %.2f cycles @2.4GHz (counter ticks/wave %.0f)\n",
	       name, waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, mean);
	fflush(stdout);
	hipFree(out); hipFree(cyc);
}

int main()
{
	for (int w : {1, 2, 4, 8}) {
		run<0>("v_fma_f32", w, 64);
		run<7>("v_mul_f32 (VOP2)", w, 64);
		run<1>("v_cndmask_b32_e64 (SGPR mask)", w, 64);
		run<3>("v_cmp_lt_f32_e64 -> SGPR", w, 64);
		run<2>("v_add_f32_dpp quad_perm", w, 64);
		run<4>("v_pk_fma_f32", w, 64);
		run<5>("v_rcp_f32", w, 64);
		run<6>("v_ldexp_f32", w, 64);
		run<8>("v_mul_f32 + s_add_u32 1:1", w, 64);
		run<9>("v_fma_f32 exec = low half", w, 64);
		run<10>("v_fma_f32 exec = 16 of each half", w, 64);
		run<11>("v_fma_f32 exec = lanes 0-15", w, 64);
		run<12>("v_fma_f32 exec = lanes 0 and 32", w, 64);
		run<13>("v_exp_f32", w, 64);
		run<14>("v_fma_f32 (SGPR operand)", w, 64);
		run<15>("v_cmp -> vcc + v_cndmask vcc", w, 64);
		run<16>("v_readlane_b32 (SGPR lane)", w, 64);
	}
	return 0;
}
