// Micro-benchmarks that put numbers on the two on-chip ceilings of the panel crossprod kernel
// (kernels_mult_pbc.hip), independent of the kernel itself:
//
//  A. stage: how fast a CU takes row panels of Y (128 rows x 64 dense columns = 64 KiB) into LDS by
//     LDS-DMA when 16 workgroups of an XCD pull the same panels, as a function of the SOURCE layout:
//       layout 0  column-major Y, leading dimension ld (pieces of 1 KiB, 8*ld bytes apart)
//       layout 1  panel-major copy Yp[panel][dense column][128] (a workgroup's 64 pieces are contiguous)
//     and of the number of wavefronts that issue (16 x 4 pieces or 8 x 8 pieces).
//  B. work: the record work alone -- LDS address add, ds_read_b64 (lane = dense column), register-index
//     switch, v_fma_f64 on an indexed accumulator -- on LDS-resident data with constant records in SGPRs:
//     no staging, no record loads, no barriers.  Variants drop one leg at a time.
//
// hipcc -O3 --offload-arch=gfx950 -o ceiling_bench ceiling_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------- A
// skew: block b starts `skew * b` panels into its row split and wraps around (the 16 blocks that
//       share panels then ask for them at different times instead of hammering the same lines)
// spread: 0 = the 16 column blocks of a (row split, dense tile) sit on one XCD (as the product kernel
//       places them); 1 = they are dealt over all 8 XCDs, two each
template <int WPB, int AUX>
__global__ void __launch_bounds__(WPB * 64)
stage_kernel(const double *__restrict__ Y, int layout, int64_t ld, int nblocks, int kt,
	     int64_t npanels, int64_t panels_per_split, int K, double *sink, int skew, int spread)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = 64 / WPB;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	int kh, sp, b;
	if (spread == 0) {
		const int xcd = L % 8, j = L / 8;
		const int u = j / nblocks;
		b = j % nblocks; kh = u % kt; sp = (u / kt) * 8 + xcd;
	} else {
		b = L % nblocks;                    // consecutive launch indices = consecutive XCDs
		const int u = L / nblocks;
		kh = u % kt; sp = u / kt;
	}
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	double acc = 0.0;
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			const double *src = layout == 0
				? Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2
				: Y + (p * K + k0 + kk) * 128 + lane * 2;
			double *dst = lds + buf * BUF + kk * RS;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
							 (__attribute__((address_space(3))) void *) dst, 16, 0, AUX);
		}
	};
	const int64_t np = pb - pa;
	const int64_t s0 = ((int64_t) skew * b) % np;
	auto panel = [&](int64_t i) { const int64_t q = i + s0; return pa + (q >= np ? q - np : q); };
	issue(panel(0), 0);
	for (int64_t i = 0; i < np; i++) {
		const int buf = (int) (i & 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (i + 1 < np) issue(panel(i + 1), buf ^ 1);
		acc += lds[buf * BUF + lane * RS + (w & 7)];
	}
	if (acc == 123.456) sink[0] = acc;
}

template <int WPB, int AUX>
static void launch_stage(int nwg, size_t ldsb, const double *Y, int layout, int64_t ld, int nblocks, int kt,
			 int64_t npanels, int64_t pps, int K, double *sink, int skew, int spread)
{
	CHECK(hipFuncSetAttribute((const void *) stage_kernel<WPB, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
	hipLaunchKernelGGL((stage_kernel<WPB, AUX>), dim3(nwg), dim3(WPB * 64), ldsb, 0, Y, layout, ld, nblocks, kt,
			   npanels, pps, K, sink, skew, spread);
}

static void run_stage(const double *Y, double *sink, int wpb, int layout, int64_t nrow, int64_t ld, const char *what,
		      int skew = 0, int spread = 0, int aux = 0)
{
	const int K = 128, kt = 2, nblocks = 16, nsplit = 8;
	const int64_t npanels = nrow / 128;
	const int64_t pps = (npanels + nsplit - 1) / nsplit;
	const int nwg = nblocks * kt * nsplit;
	const size_t ldsb = (size_t) 2 * 64 * 129 * 8;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 6; rep++) {
		CHECK(hipEventRecord(e0));
#define LS(W, A) launch_stage<W, A>(nwg, ldsb, Y, layout, ld, nblocks, kt, npanels, pps, K, sink, skew, spread)
		if (wpb == 8) LS(8, 0);
		else if (aux == 1) LS(16, 1);
		else if (aux == 2) LS(16, 2);
		else if (aux == 3) LS(16, 3);
		else LS(16, 0);
#undef LS
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best) best = ms;
	}
	const double bytes = (double) nblocks * npanels * 128 * K * 8;
	printf("stage %-34s skew %d spread %d aux %d wpb %2d: %.3f ms, %.1f GB staged, %.1f TB/s, %.1f B/clk/CU, %.0f cycles/panel\n", what, skew, spread, aux, wpb,
	       best, bytes / 1e9, bytes / best / 1e9, bytes / 256 / (best * 1e-3 * 2.4e9), best * 1e-3 * 2.4e9 / pps);
	CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

// A2. stage2: the same staging with (rot) the piece order rotated by the column block, so that the 16
// workgroups that share a panel do not ask for the same lines at the same moment, and (mode) the queue
// kept busy: 0 = as above (wait for the panel, barrier, issue the next), 1 = the next panel is issued
// before the wait (two panels in flight, counted vmcnt), barrier per panel, 2 = the same without any
// barrier (every wavefront streams its own pieces).  Nothing reads the buffers: an upper bound on what
// two buffers can take in.
template <int WPB>
__global__ void __launch_bounds__(WPB * 64)
stage2_kernel(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	      int64_t panels_per_split, double *sink, int rot, int mode)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = 64 / WPB;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8, u = j / nblocks;
	const int b = j % nblocks, kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	const int wr = rot ? (w + b * rot) % WPB : w;
	double acc = 0.0;
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = wr * NPIECE + q;
			const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
			double *dst = lds + buf * BUF + kk * RS;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
							 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
		}
	};
	const int64_t np = pb - pa;
	issue(pa, 0);
	for (int64_t i = 0; i < np; i++) {
		const int buf = (int) (i & 1);
		if (mode == 0) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			if (i + 1 < np) issue(pa + i + 1, buf ^ 1);
		} else {
			if (i + 1 < np) {
				issue(pa + i + 1, buf ^ 1);
				if (NPIECE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
				else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
			} else {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
			if (mode == 1) __builtin_amdgcn_s_barrier();
		}
		if ((i & 63) == 63) acc += lds[buf * BUF + lane * RS + (w & 7)];
	}
	if (acc == 123.456) sink[0] = acc;
}

static void run_stage2(const double *Y, double *sink, int wpb, int rot, int mode)
{
	const int K = 128, kt = 2, nblocks = 16, nsplit = 8;
	const int64_t nrow = 999936, ld = 1000000, npanels = nrow / 128;
	const int64_t pps = (npanels + nsplit - 1) / nsplit;
	const int nwg = nblocks * kt * nsplit;
	const size_t ldsb = (size_t) 2 * 64 * 129 * 8;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 6; rep++) {
		CHECK(hipEventRecord(e0));
		if (wpb == 16) {
			CHECK(hipFuncSetAttribute((const void *) stage2_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
			hipLaunchKernelGGL((stage2_kernel<16>), dim3(nwg), dim3(16 * 64), ldsb, 0, Y, ld, nblocks, kt, npanels, pps, sink, rot, mode);
		} else {
			CHECK(hipFuncSetAttribute((const void *) stage2_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
			hipLaunchKernelGGL((stage2_kernel<8>), dim3(nwg), dim3(8 * 64), ldsb, 0, Y, ld, nblocks, kt, npanels, pps, sink, rot, mode);
		}
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best) best = ms;
	}
	const double bytes = (double) nblocks * npanels * 128 * K * 8;
	printf("stage2 wpb %2d rot %d mode %d (%s): %.3f ms, %.1f TB/s, %.1f GB/s per CU, %.0f ns/panel\n", wpb, rot, mode,
	       mode == 0 ? "drain + barrier" : mode == 1 ? "2 panels in flight + barrier" : "2 panels in flight, no barrier",
	       best, bytes / best / 1e9, bytes / 256 / best / 1e6, best * 1e6 / pps);
	CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

// ------------------------------------------------------------------------------------------- B
typedef double d16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// records: meta words in s[28:35] (row byte offset << 16 | 2 * column); y sets v[12:27] and v[108:123];
// lane base %[lb]; accumulators v[44:107] (register-indexed: v[44 + 2 * column])
#define A1(Y, M) "v_add_u32_sdwa v" #Y ", s" #M ", %[lb] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
#define P1(Y, M) "v_add_u32 v" #Y ", s" #M ", %[lb]\n\t"
#define R1(Y, Y1) "ds_read_b64 v[" #Y ":" #Y1 "], v" #Y "\n\t"
#define R2(Y, Y3) "ds_read_b128 v[" #Y ":" #Y3 "], v" #Y "\n\t"
#define ION(M) "s_set_gpr_idx_on s" #M ", gpr_idx(SRC2,DST)\n\t"
#define I1(M) "s_set_gpr_idx_idx s" #M "\n\t"
#define IOFF "s_set_gpr_idx_off\n\t"
#define F1(Y, Y1) "v_fma_f64 v[44:45], %[one], v[" #Y ":" #Y1 "], v[44:45]\n\t"
#define F2(Y, Y1, Y2, Y3) F1(Y, Y1) "v_fma_f64 v[76:77], %[one], v[" #Y2 ":" #Y3 "], v[76:77]\n\t"
#define ADDR_A A1(12, 28) A1(14, 29) A1(16, 30) A1(18, 31) A1(20, 32) A1(22, 33) A1(24, 34) A1(26, 35)
#define ADDR_B A1(108, 28) A1(110, 29) A1(112, 30) A1(114, 31) A1(116, 32) A1(118, 33) A1(120, 34) A1(122, 35)
#define PADDR_A P1(12, 28) P1(14, 29) P1(16, 30) P1(18, 31) P1(20, 32) P1(22, 33) P1(24, 34) P1(26, 35)
#define PADDR_B P1(108, 28) P1(110, 29) P1(112, 30) P1(114, 31) P1(116, 32) P1(118, 33) P1(120, 34) P1(122, 35)
#define READ_A R1(12, 13) R1(14, 15) R1(16, 17) R1(18, 19) R1(20, 21) R1(22, 23) R1(24, 25) R1(26, 27)
#define READ_B R1(108, 109) R1(110, 111) R1(112, 113) R1(114, 115) R1(116, 117) R1(118, 119) R1(120, 121) R1(122, 123)
#define FMAI_A ION(28) F1(12, 13) I1(29) F1(14, 15) I1(30) F1(16, 17) I1(31) F1(18, 19) I1(32) F1(20, 21) I1(33) F1(22, 23) I1(34) F1(24, 25) I1(35) F1(26, 27) IOFF
#define FMAI_B ION(28) F1(108, 109) I1(29) F1(110, 111) I1(30) F1(112, 113) I1(31) F1(114, 115) I1(32) F1(116, 117) I1(33) F1(118, 119) I1(34) F1(120, 121) I1(35) F1(122, 123) IOFF
// fixed accumulators (no index mode): 8 different registers so that the FMAs stay independent
#define G1(D, D1, Y, Y1) "v_fma_f64 v[" #D ":" #D1 "], %[one], v[" #Y ":" #Y1 "], v[" #D ":" #D1 "]\n\t"
#define FMAF_A G1(44, 45, 12, 13) G1(54, 55, 14, 15) G1(62, 63, 16, 17) G1(68, 69, 18, 19) G1(80, 81, 20, 21) G1(88, 89, 22, 23) G1(98, 99, 24, 25) G1(106, 107, 26, 27)
#define FMAF_B G1(44, 45, 108, 109) G1(54, 55, 110, 111) G1(62, 63, 112, 113) G1(68, 69, 114, 115) G1(80, 81, 116, 117) G1(88, 89, 118, 119) G1(98, 99, 120, 121) G1(106, 107, 122, 123)
#define LOOP_HEAD "s_mov_b32 s36, m0\n1:\n\t"
#define LOOP_TAIL "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_mov_b32 m0, s36\n\t"
#define WAIT "s_waitcnt lgkmcnt(0)\n\t"
#define OPS : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya), "+{v[108:123]}"(yb) \
	    : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36"

// MODE 0: add + read + idx + fma (two y sets: the reads of one batch fly under the FMAs of the other)
// MODE 1: no index switch (fixed accumulators)    MODE 2: no LDS read    MODE 3: idx + fma only
// MODE 4: 2 dense columns per lane: ds_read_b128 + idx + 2 fma per record   MODE 5: fma only, fixed acc
template <int MODE>
__global__ void __launch_bounds__(1024)
work_kernel(double *sink, int iters, int lds_doubles)
{
	extern __shared__ double lds[];
	const int tid = threadIdx.x, lane = tid & 63;
	for (int i = tid; i < lds_doubles; i += blockDim.x) lds[i] = 1.0 + 1e-9 * i;
	__syncthreads();
	d16 acc0 = 0.0, acc1 = 0.0;
	typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
	u32x16 ya = 0, yb = 0;
	const uint32_t lanebase = (uint32_t) lane * (MODE == 4 ? 2064u : 1032u);
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};          // (<= 15: MODE 4 keeps two banks of 16)
#pragma unroll
	for (int q = 0; q < 8; q++)
		meta[q] = (MODE == 6 || MODE == 7) ? (uint32_t) (rows[q] * 8)
			: ((uint32_t) (rows[q] * (MODE == 4 ? 16 : 8)) << 16) | (uint32_t) (2 * cols[q]);
	uint32_t n = (uint32_t) iters;
	const double one = 1.0000001;
	if constexpr (MODE == 0) {
		asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL OPS);
	} else if constexpr (MODE == 1) {
		asm volatile(LOOP_HEAD ADDR_A READ_A FMAF_B WAIT ADDR_B READ_B FMAF_A WAIT LOOP_TAIL OPS);
	} else if constexpr (MODE == 2) {
		asm volatile(LOOP_HEAD ADDR_A FMAI_B ADDR_B FMAI_A LOOP_TAIL OPS);
	} else if constexpr (MODE == 3) {
		asm volatile(LOOP_HEAD FMAI_B FMAI_A LOOP_TAIL OPS);
	} else if constexpr (MODE == 6) {   // plain (non-SDWA) add of a whole SGPR word; fixed accumulators
		asm volatile(LOOP_HEAD PADDR_A READ_A FMAF_B WAIT PADDR_B READ_B FMAF_A WAIT LOOP_TAIL OPS);
	} else if constexpr (MODE == 7) {   // plain add only + fma (no read)
		asm volatile(LOOP_HEAD PADDR_A FMAF_B PADDR_B FMAF_A LOOP_TAIL OPS);
	} else if constexpr (MODE == 8) {   // SDWA add + fma, fixed acc (no read, no idx)
		asm volatile(LOOP_HEAD ADDR_A FMAF_B ADDR_B FMAF_A LOOP_TAIL OPS);
	} else if constexpr (MODE == 5) {
		asm volatile(LOOP_HEAD FMAF_B FMAF_A LOOP_TAIL OPS);
	} else {
		// 8 records per half trip as well, 4 + 4: set a = v[12:27] as 4 quads, set b = v[108:123]
		asm volatile(LOOP_HEAD
			     A1(12, 28) A1(16, 29) A1(20, 30) A1(24, 31) R2(12, 15) R2(16, 19) R2(20, 23) R2(24, 27)
			     ION(32) F2(108, 109, 110, 111) I1(33) F2(112, 113, 114, 115) I1(34) F2(116, 117, 118, 119) I1(35) F2(120, 121, 122, 123) IOFF
			     WAIT
			     A1(108, 32) A1(112, 33) A1(116, 34) A1(120, 35) R2(108, 111) R2(112, 115) R2(116, 119) R2(120, 123)
			     ION(28) F2(12, 13, 14, 15) I1(29) F2(16, 17, 18, 19) I1(30) F2(20, 21, 22, 23) I1(31) F2(24, 25, 26, 27) IOFF
			     WAIT LOOP_TAIL OPS);
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] == 123.456) sink[0] = acc0[1];
}


// ------------------------------------------------------------------------------------------- C
// combo: the panel loop of the product kernel without its record loads -- per 128-row panel one barrier,
// the LDS-DMA pieces of the next panel issued right behind it, then this wavefront's share of the
// record work on constant records (trips of 16 records, mode 0 above).  Compares workgroup shapes at
// the same register budget per CU: 16 wavefronts x 40 columns (128 VGPRs), 12 x 70 (168), 8 x 106 (256).
// YS1: one y set (the refill read of y_j goes out right behind the FMA that used it), as a 12-wavefront
// build would need.
#define R1A(Y, Y1, A) "ds_read_b64 v[" #Y ":" #Y1 "], v" #A "\n\t"
#define S1(A, M) "v_add_u32_sdwa v" #A ", s" #M ", %[lb] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
// one y set: addresses v[2:9], y v[12:27]
#define ADDR_1 S1(2, 28) S1(3, 29) S1(4, 30) S1(5, 31) S1(6, 32) S1(7, 33) S1(8, 34) S1(9, 35)
#define FR(M, Y, Y1, A) I1(M) F1(Y, Y1) R1A(Y, Y1, A)
#define FMAR_1 ION(28) F1(12, 13) R1A(12, 13, 2) FR(29, 14, 15, 3) FR(30, 16, 17, 4) FR(31, 18, 19, 5) FR(32, 20, 21, 6) FR(33, 22, 23, 7) FR(34, 24, 25, 8) FR(35, 26, 27, 9) IOFF
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(16 * 64)
combo_16(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	     int64_t panels_per_split, double *sink, double trips_per_panel, int dma, int work)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = (64 + 16 - 1) / 16;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	for (int i = tid; i < 2 * BUF; i += 16 * 64) lds[i] = 1.0;
	__syncthreads();
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			if (kk < 64) {
				const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
				double *dst = lds + buf * BUF + kk * RS;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
								 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
			}
		}
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0, hi = 0;
	u32x8 ad = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	if (dma) issue(pa, 0);
	for (int64_t p = pa; p < pb; p++) {
		const int buf = (int) ((p - pa) & 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (dma && p + 1 < pb) issue(p + 1, buf ^ 1);
		const int64_t i = p - pa;
		uint32_t n = (uint32_t) ((int64_t) ((i + 1) * trips_per_panel) - (int64_t) (i * trips_per_panel));
		n = __builtin_amdgcn_readfirstlane(n);
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		if (work && n > 0) {
			if constexpr (false) {
				asm volatile(LOOP_HEAD ADDR_1 FMAR_1 WAIT ADDR_1 FMAR_1 WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[2:9]}"(ad), "+{v[112:127]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			} else {
				asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[108:123]}"(yb), "+{v[112:127]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			}
		}
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] + (double) hi[2] + (double) ad[1] == 123.456) sink[0] = acc0[1];
}
// combo_16 with the DMA pieces issued by the two oldest wavefronts of every SIMD only (8 pieces each): they wait
// at the barrier anyway (`trace`), the youngest are the panel's critical path
__global__ void __launch_bounds__(16 * 64)
combo_16old(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	     int64_t panels_per_split, double *sink, double trips_per_panel, int dma, int work)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = (64 + 16 - 1) / 16;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	for (int i = tid; i < 2 * BUF; i += 16 * 64) lds[i] = 1.0;
	__syncthreads();
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < 2 * NPIECE; q++) {          // the two oldest wavefronts of every SIMD stage the whole panel
			const int kk = w * 2 * NPIECE + q;
			if (w < 8 && kk < 64) {
				const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
				double *dst = lds + buf * BUF + kk * RS;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
								 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
			}
		}
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0, hi = 0;
	u32x8 ad = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	if (dma) issue(pa, 0);
	for (int64_t p = pa; p < pb; p++) {
		const int buf = (int) ((p - pa) & 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (dma && p + 1 < pb) issue(p + 1, buf ^ 1);
		const int64_t i = p - pa;
		uint32_t n = (uint32_t) ((int64_t) ((i + 1) * trips_per_panel) - (int64_t) (i * trips_per_panel));
		n = __builtin_amdgcn_readfirstlane(n);
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		if (work && n > 0) {
			if constexpr (false) {
				asm volatile(LOOP_HEAD ADDR_1 FMAR_1 WAIT ADDR_1 FMAR_1 WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[2:9]}"(ad), "+{v[112:127]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			} else {
				asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[108:123]}"(yb), "+{v[112:127]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			}
		}
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] + (double) hi[2] + (double) ad[1] == 123.456) sink[0] = acc0[1];
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define ADDR_A10 ADDR_A A1(28, 36) A1(30, 37)
#define ADDR_B10 ADDR_B A1(124, 36) A1(126, 37)
#define READ_A10 READ_A R1(28, 29) R1(30, 31)
#define READ_B10 READ_B R1(124, 125) R1(126, 127)
#define FMAI_A10 ION(28) F1(12, 13) I1(29) F1(14, 15) I1(30) F1(16, 17) I1(31) F1(18, 19) I1(32) F1(20, 21) I1(33) F1(22, 23) I1(34) F1(24, 25) I1(35) F1(26, 27) I1(36) F1(28, 29) I1(37) F1(30, 31) IOFF
#define FMAI_B10 ION(28) F1(108, 109) I1(29) F1(110, 111) I1(30) F1(112, 113) I1(31) F1(114, 115) I1(32) F1(116, 117) I1(33) F1(118, 119) I1(34) F1(120, 121) I1(35) F1(122, 123) I1(36) F1(124, 125) I1(37) F1(126, 127) IOFF
// combo_16 with 10 records per batch (10 meta words, two y sets of 20 registers)
__global__ void __launch_bounds__(16 * 64)
combo_16x10(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	     int64_t panels_per_split, double *sink, double trips_per_panel, int dma, int work)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = (64 + 16 - 1) / 16;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	for (int i = tid; i < 2 * BUF; i += 16 * 64) lds[i] = 1.0;
	__syncthreads();
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			if (kk < 64) {
				const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
				double *dst = lds + buf * BUF + kk * RS;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
								 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
			}
		}
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0, hi = 0;
	u32x4 ya4 = 0, yb4 = 0;
	const uint32_t m8 = ((uint32_t) (25 * 8) << 16) | 8u, m9 = ((uint32_t) (111 * 8) << 16) | 22u;
	u32x8 ad = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	if (dma) issue(pa, 0);
	for (int64_t p = pa; p < pb; p++) {
		const int buf = (int) ((p - pa) & 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (dma && p + 1 < pb) issue(p + 1, buf ^ 1);
		const int64_t i = p - pa;
		uint32_t n = (uint32_t) ((int64_t) ((i + 1) * trips_per_panel) - (int64_t) (i * trips_per_panel));
		n = __builtin_amdgcn_readfirstlane(n);
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		if (work && n > 0) {
			if constexpr (false) {
				asm volatile(LOOP_HEAD ADDR_1 FMAR_1 WAIT ADDR_1 FMAR_1 WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[2:9]}"(ad), "+{v[112:127]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			} else {
				asm volatile("s_mov_b32 s40, m0\n1:\n\t" ADDR_A10 READ_A10 FMAI_B10 WAIT ADDR_B10 READ_B10 FMAI_A10 WAIT
					     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_mov_b32 m0, s40\n\t"
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya), "+{v[28:31]}"(ya4),
					       "+{v[108:123]}"(yb), "+{v[124:127]}"(yb4)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta), "{s36}"(m8), "{s37}"(m9) : "memory", "scc", "s40");
			}
		}
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] + (double) hi[2] + (double) ad[1] + (double) ya4[3] + (double) yb4[1] == 123.456) sink[0] = acc0[1];
}
// combo_16 with the address adds of the next reads AHEAD of the wait for the current ones
__global__ void __launch_bounds__(16 * 64)
combo_16r(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	     int64_t panels_per_split, double *sink, double trips_per_panel, int dma, int work)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = (64 + 16 - 1) / 16;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	for (int i = tid; i < 2 * BUF; i += 16 * 64) lds[i] = 1.0;
	__syncthreads();
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			if (kk < 64) {
				const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
				double *dst = lds + buf * BUF + kk * RS;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
								 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
			}
		}
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0, hi = 0;
	u32x8 ad = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	if (dma) issue(pa, 0);
	for (int64_t p = pa; p < pb; p++) {
		const int buf = (int) ((p - pa) & 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (dma && p + 1 < pb) issue(p + 1, buf ^ 1);
		const int64_t i = p - pa;
		uint32_t n = (uint32_t) ((int64_t) ((i + 1) * trips_per_panel) - (int64_t) (i * trips_per_panel));
		n = __builtin_amdgcn_readfirstlane(n);
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		if (work && n > 0) {
			if constexpr (false) {
				asm volatile(LOOP_HEAD ADDR_1 FMAR_1 WAIT ADDR_1 FMAR_1 WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[2:9]}"(ad), "+{v[112:127]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			} else {
				asm volatile("s_mov_b32 s36, m0\n\t" ADDR_A "1:\n\t" READ_A FMAI_B ADDR_B WAIT READ_B FMAI_A ADDR_A WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[108:123]}"(yb), "+{v[112:127]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			}
		}
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] + (double) hi[2] + (double) ad[1] == 123.456) sink[0] = acc0[1];
}
__global__ void __launch_bounds__(12 * 64)
combo_12(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	     int64_t panels_per_split, double *sink, double trips_per_panel, int dma, int work)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = (64 + 12 - 1) / 12;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	for (int i = tid; i < 2 * BUF; i += 12 * 64) lds[i] = 1.0;
	__syncthreads();
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			if (kk < 64) {
				const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
				double *dst = lds + buf * BUF + kk * RS;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
								 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
			}
		}
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0, hi = 0;
	u32x8 ad = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	if (dma) issue(pa, 0);
	for (int64_t p = pa; p < pb; p++) {
		const int buf = (int) ((p - pa) & 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (dma && p + 1 < pb) issue(p + 1, buf ^ 1);
		const int64_t i = p - pa;
		uint32_t n = (uint32_t) ((int64_t) ((i + 1) * trips_per_panel) - (int64_t) (i * trips_per_panel));
		n = __builtin_amdgcn_readfirstlane(n);
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		if (work && n > 0) {
			if constexpr (true) {
				asm volatile(LOOP_HEAD ADDR_1 FMAR_1 WAIT ADDR_1 FMAR_1 WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[2:9]}"(ad), "+{v[152:167]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			} else {
				asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[108:123]}"(yb), "+{v[152:167]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			}
		}
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] + (double) hi[2] + (double) ad[1] == 123.456) sink[0] = acc0[1];
}
__global__ void __launch_bounds__(8 * 64)
combo_8(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	     int64_t panels_per_split, double *sink, double trips_per_panel, int dma, int work)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = (64 + 8 - 1) / 8;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	for (int i = tid; i < 2 * BUF; i += 8 * 64) lds[i] = 1.0;
	__syncthreads();
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			if (kk < 64) {
				const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
				double *dst = lds + buf * BUF + kk * RS;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
								 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
			}
		}
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0, hi = 0;
	u32x8 ad = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	if (dma) issue(pa, 0);
	for (int64_t p = pa; p < pb; p++) {
		const int buf = (int) ((p - pa) & 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (dma && p + 1 < pb) issue(p + 1, buf ^ 1);
		const int64_t i = p - pa;
		uint32_t n = (uint32_t) ((int64_t) ((i + 1) * trips_per_panel) - (int64_t) (i * trips_per_panel));
		n = __builtin_amdgcn_readfirstlane(n);
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		if (work && n > 0) {
			if constexpr (false) {
				asm volatile(LOOP_HEAD ADDR_1 FMAR_1 WAIT ADDR_1 FMAR_1 WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[2:9]}"(ad), "+{v[240:255]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			} else {
				asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[108:123]}"(yb), "+{v[240:255]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			}
		}
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] + (double) hi[2] + (double) ad[1] == 123.456) sink[0] = acc0[1];
}
__global__ void __launch_bounds__(8 * 64)
combo_8y1(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	     int64_t panels_per_split, double *sink, double trips_per_panel, int dma, int work)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = (64 + 8 - 1) / 8;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	for (int i = tid; i < 2 * BUF; i += 8 * 64) lds[i] = 1.0;
	__syncthreads();
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			if (kk < 64) {
				const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
				double *dst = lds + buf * BUF + kk * RS;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
								 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
			}
		}
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0, hi = 0;
	u32x8 ad = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	if (dma) issue(pa, 0);
	for (int64_t p = pa; p < pb; p++) {
		const int buf = (int) ((p - pa) & 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (dma && p + 1 < pb) issue(p + 1, buf ^ 1);
		const int64_t i = p - pa;
		uint32_t n = (uint32_t) ((int64_t) ((i + 1) * trips_per_panel) - (int64_t) (i * trips_per_panel));
		n = __builtin_amdgcn_readfirstlane(n);
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		if (work && n > 0) {
			if constexpr (true) {
				asm volatile(LOOP_HEAD ADDR_1 FMAR_1 WAIT ADDR_1 FMAR_1 WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[2:9]}"(ad), "+{v[240:255]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			} else {
				asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[108:123]}"(yb), "+{v[240:255]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			}
		}
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] + (double) hi[2] + (double) ad[1] == 123.456) sink[0] = acc0[1];
}
__global__ void __launch_bounds__(16 * 64)
combo_16y1(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	     int64_t panels_per_split, double *sink, double trips_per_panel, int dma, int work)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = (64 + 16 - 1) / 16;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	for (int i = tid; i < 2 * BUF; i += 16 * 64) lds[i] = 1.0;
	__syncthreads();
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			if (kk < 64) {
				const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
				double *dst = lds + buf * BUF + kk * RS;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
								 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
			}
		}
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0, hi = 0;
	u32x8 ad = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	if (dma) issue(pa, 0);
	for (int64_t p = pa; p < pb; p++) {
		const int buf = (int) ((p - pa) & 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (dma && p + 1 < pb) issue(p + 1, buf ^ 1);
		const int64_t i = p - pa;
		uint32_t n = (uint32_t) ((int64_t) ((i + 1) * trips_per_panel) - (int64_t) (i * trips_per_panel));
		n = __builtin_amdgcn_readfirstlane(n);
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		if (work && n > 0) {
			if constexpr (true) {
				asm volatile(LOOP_HEAD ADDR_1 FMAR_1 WAIT ADDR_1 FMAR_1 WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[2:9]}"(ad), "+{v[112:127]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			} else {
				asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(n), "+{v[12:27]}"(ya),
					       "+{v[108:123]}"(yb), "+{v[112:127]}"(hi)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			}
		}
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] + (double) hi[2] + (double) ad[1] == 123.456) sink[0] = acc0[1];
}

typedef void (*combo_fn)(const double *, int64_t, int, int, int64_t, int64_t, double *, double, int, int);
static void run_combo(const double *Y, double *sink, combo_fn fn, int wpb, int nblocks, int nsplit, double rec_per_panel,
		      int dma, int work, const char *what, int mult = 1)
{
	// mult > 1: a barrier every `mult` panels' worth of records (same total work; meaningful with dma = 0 only)
	const int kt = 2;
	rec_per_panel *= mult;
	const int64_t nrow = 999936, ld = 1000000, npanels = nrow / 128 / mult;
	const int64_t pps = (npanels + nsplit - 1) / nsplit;
	const int nwg = nblocks * kt * nsplit;
	const size_t ldsb = (size_t) 2 * 64 * 129 * 8;
	CHECK(hipFuncSetAttribute((const void *) fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float best = 1e30f;
	const double trips = rec_per_panel / wpb / 16.0;
	for (int rep = 0; rep < 4; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(fn, dim3(nwg), dim3(wpb * 64), ldsb, 0, Y, ld, nblocks, kt, npanels, pps, sink, trips, dma, work);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best) best = ms;
	}
	printf("combo %-26s %2d waves, %2d column blocks x %2d row splits = %3d WGs, %.0f records/WG-panel, dma %d work %d: %.3f ms (%.0f cycles/panel)\n",
	       what, wpb, nblocks, nsplit, nwg, rec_per_panel, dma, work, best, best * 1e-3 * 2.4e9 / pps);
	CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}


// ------------------------------------------------------------------------------------------- D
// flag: the combo loop with the workgroup barrier replaced by two counters per panel buffer in LDS --
// landed[buf] (a wavefront's DMA pieces of the panel are in LDS) and done[buf] (a wavefront has finished
// reading it).  A wavefront starts panel i as soon as all pieces of i have landed, issues its pieces of
// i + 1 at the first trip boundary at which every wavefront is done with i - 1, and publishes them two
// trips later.  Slack: one panel.  imbalance: the wavefronts' trip counts per panel vary like the
// record counts of real tiles (Poisson(51) per 40 columns x 128 rows); barrier = 1 gives the same loop
// with s_barrier for comparison.
__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <bool barrier>
__global__ void __launch_bounds__(16 * 64)
combo_flag(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	   int64_t panels_per_split, double *sink, int imbalance)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS, NPIECE = 4;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	uint32_t *flags = (uint32_t *) (lds + 2 * BUF);          // landed[0], landed[1], done[0], done[1]
	for (int i = tid; i < 2 * BUF; i += 16 * 64) lds[i] = 1.0;
	if (tid < 4) flags[tid] = 0;
	__syncthreads();
	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			const double *src = Y + (int64_t) (k0 + kk) * ld + p * 128 + lane * 2;
			double *dst = lds + buf * BUF + kk * RS;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
							 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
		}
	};
	// (inline asm: for a C++ LDS access the compiler puts s_waitcnt vmcnt(0) first -- the LDS-DMA in flight might
	// alias it -- and every poll would wait for the whole panel to land; round 2's numbers for this protocol had that)
	const uint32_t flags_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint32_t *) flags;
	auto peek = [&](int idx) {
		uint32_t v;
		asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flags_lds + 4u * (uint32_t) idx) : "memory");
		return (uint32_t) __builtin_amdgcn_readfirstlane(v);
	};
	auto bump = [&](int idx) {
		if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(flags_lds + 4u * (uint32_t) idx), "v"(1u) : "memory");
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	const int64_t np = pb - pa;
	issue(pa, 0);
	int carry = 0;                                          // records not yet turned into whole trips of 16
	for (int64_t i = 0; i < np; i++) {
		const int buf = (int) (i & 1);
		// records of this wavefront in this panel: 51.2 on average
		int recs = 51 + (int) ((i * 16 + w) % 5 == 0);
		if (imbalance) {
			const uint32_t h = mix32((uint32_t) (L * 1000003 + i * 16 + w));
			const int z = (int) (h & 15) + (int) ((h >> 4) & 15) + (int) ((h >> 8) & 15) + (int) ((h >> 12) & 15) - 30;   // ~N(0, 9.2)
			recs = 51 + (z * 7) / 9 + (int) ((i * 16 + w) % 5 == 0);
			if (recs < 8) recs = 8;
		}
		carry += recs;
		int n = __builtin_amdgcn_readfirstlane(carry / 16);
		carry -= n * 16;
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		if constexpr (barrier) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			if (i + 1 < np) issue(pa + i + 1, buf ^ 1);
			for (int t = 0; t < n; t++) {
				uint32_t one_trip = 1;
				asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL
					     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(one_trip), "+{v[12:27]}"(ya),
					       "+{v[108:123]}"(yb)
					     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
			}
		} else {
		// ---- flag protocol
		// panel i complete in LDS?  (my own pieces were published during panel i - 1, or here for the first)
		if (i == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); bump(buf); }
		const uint32_t need = 16u * (uint32_t) ((i >> 1) + 1);
		while (peek(buf) < need) __builtin_amdgcn_s_sleep(1);
		bool issued = i + 1 >= np, published = issued;
		int issued_at = 0;
		for (int t = 0; t <= n; t++) {
			if (!issued && (i == 0 || peek(2 + (buf ^ 1)) >= 16u * (uint32_t) (((i - 1) >> 1) + 1))) {
				issue(pa + i + 1, buf ^ 1);
				issued = true; issued_at = t;
			}
			if (issued && !published && t >= issued_at + 2) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				bump(buf ^ 1);
				published = true;
			}
			if (t == n) break;
			uint32_t one_trip = 1;
			asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL
				     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(one_trip), "+{v[12:27]}"(ya),
				       "+{v[108:123]}"(yb)
				     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
		}
		bump(2 + buf);                                      // done with panel i
		if (!issued) {
			while (peek(2 + (buf ^ 1)) < 16u * (uint32_t) (((i - 1) >> 1) + 1)) __builtin_amdgcn_s_sleep(1);
			issue(pa + i + 1, buf ^ 1);
		}
		if (!published) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			bump(buf ^ 1);
		}
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] == 123.456) sink[0] = acc0[1];
}

static void run_flag(const double *Y, double *sink, int imbalance, int barrier, const char *what)
{
	const int kt = 2, nblocks = 16, nsplit = 8;
	const int64_t nrow = 999936, ld = 1000000, npanels = nrow / 128;
	const int64_t pps = (npanels + nsplit - 1) / nsplit;
	const int nwg = nblocks * kt * nsplit;
	const size_t ldsb = (size_t) 2 * 64 * 129 * 8 + 64;
	CHECK(hipFuncSetAttribute((const void *) combo_flag<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
	CHECK(hipFuncSetAttribute((const void *) combo_flag<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++) {
		CHECK(hipEventRecord(e0));
		if (barrier) hipLaunchKernelGGL(combo_flag<true>, dim3(nwg), dim3(1024), ldsb, 0, Y, ld, nblocks, kt, npanels, pps, sink, imbalance);
		else hipLaunchKernelGGL(combo_flag<false>, dim3(nwg), dim3(1024), ldsb, 0, Y, ld, nblocks, kt, npanels, pps, sink, imbalance);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best) best = ms;
	}
	printf("flag  %-40s imbalance %d: %.3f ms (%.0f cycles/panel)\n", what, imbalance, best, best * 1e-3 * 2.4e9 / pps);
	CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}



// ------------------------------------------------------------------------------------------- F
// ring: the combo loop on a ring of B panels of H rows per dense column (LDS image [64][B * H + 1]) with the
// workgroup barrier replaced by two counters per ring slot -- landed[slot] (a wavefront's DMA pieces of the
// panel are in LDS) and done[slot] (a wavefront has finished reading it).  A wavefront may run up to B - 1
// panels ahead of the slowest: it starts panel i when all pieces of i have landed, issues its pieces of
// panel i + B - 1 (into the slot of panel i - 1) at the first trip boundary at which everybody is done with
// i - 1, and publishes them a panel later (counted vmcnt: the younger pieces keep flying).  Record counts as
// in `flag` (Poisson-like around 51.2 * H / 128 per wavefront and panel).
template <int B, int H>
__global__ void __launch_bounds__(16 * 64)
combo_ring(const double *__restrict__ Y, int64_t ld, int nblocks, int kt, int64_t npanels,
	   int64_t panels_per_split, double *sink, int imbalance)
{
	extern __shared__ double lds[];
	constexpr int RS = B * H + 1, NPIECE = 4, D = B - 1;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels) pb = npanels;
	if (pa >= pb) return;
	const int k0 = kh * 64;
	uint32_t *flags = (uint32_t *) (lds + 64 * RS);        // landed[B], done[B]
	for (int i = tid; i < 64 * RS; i += 16 * 64) lds[i] = 1.0;
	if (tid < 2 * B) flags[tid] = 0;
	__syncthreads();
	auto issue = [&](int64_t p, int slot) {
		if (lane < H / 2) {
#pragma unroll
			for (int q = 0; q < NPIECE; q++) {
				const int kk = w * NPIECE + q;
				const double *src = Y + (int64_t) (k0 + kk) * ld + p * H + lane * 2;
				double *dst = lds + kk * RS + slot * H;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
								 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
			}
		}
	};
	// (inline asm: for a C++ LDS access the compiler puts s_waitcnt vmcnt(0) first -- the LDS-DMA in flight might
	// alias it -- and every poll would wait for the whole panel to land; round 2's numbers for this protocol had that)
	const uint32_t flags_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint32_t *) flags;
	auto peek = [&](int idx) {
		uint32_t v;
		asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flags_lds + 4u * (uint32_t) idx) : "memory");
		return (uint32_t) __builtin_amdgcn_readfirstlane(v);
	};
	auto bump = [&](int idx) {
		if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(flags_lds + 4u * (uint32_t) idx), "v"(1u) : "memory");
	};
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0;
	u32x8 meta;
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) ((((q * 2 + 1) * H) / 17) * 8) << 16) | (uint32_t) (2 * cols[q]);   // rows inside the panel
	const double one = 1.0000001;
	const int64_t np = pb - pa;
	// fill: panels 0 .. D - 1
	for (int i = 0; i < D && i < np; i++) issue(pa + i, i);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	for (int i = 0; i < D && i < np; i++) bump(i);
	// (integer bookkeeping only: the loop below runs once per H rows, and whatever it costs beyond the
	// protocol itself would be charged to the protocol)
	int carry = 0;
	constexpr int MEAN10 = 512 * H / 128;                   // records per wavefront and panel, tenths
	int pend_slot = -1;
	int slot = 0, tslot = D % B;
	uint32_t need = 16u, need_done = 0u;                    // counters' targets for panel i / for panel i - 1 done
	int acc10 = 0;
	for (int64_t i = 0; i < np; i++) {
		acc10 += MEAN10;
		int recs = acc10 / 10; acc10 -= recs * 10;
		if (imbalance) {
			const uint32_t h = mix32((uint32_t) (L * 1000003 + (int) i * 16 + w));
			const int z = (int) (h & 15) + (int) ((h >> 4) & 15) + (int) ((h >> 8) & 15) + (int) ((h >> 12) & 15) - 30;   // ~N(0, 9.2)
			// sd of a Poisson count with this mean: sqrt(51.2 H / 128) = 7.155 sqrt(H / 128); z has sd 9.2
			constexpr int K256 = (int) (256.0 * 7.155 / 9.2 * (H == 128 ? 1.0 : H == 104 ? 0.9014 : H == 72 ? 0.75 : H == 64 ? 0.7071 : H == 60 ? 0.6847 : 0.6124));
			recs += (z * K256) >> 8;
			if (recs < 2) recs = 2;
		}
		carry += recs;
		int n = __builtin_amdgcn_readfirstlane(carry >> 4);
		carry &= 15;
		const uint32_t lanebase = (uint32_t) lane * (RS * 8u) + (uint32_t) slot * (H * 8u);
		while (peek(slot) < need) __builtin_amdgcn_s_sleep(1);
		bool issued = i + D >= np;
		const bool had_issue = !issued;
		for (int t = 0; t <= n; t++) {
			if (!issued && (i == 0 || peek(B + tslot) >= need_done)) {
				issue(pa + i + D, tslot);
				issued = true;
			}
			if (t == n) break;
			uint32_t one_trip = 1;
			asm volatile(LOOP_HEAD ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT LOOP_TAIL
				     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(one_trip), "+{v[12:27]}"(ya),
				       "+{v[108:123]}"(yb)
				     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36");
		}
		bump(B + slot);                                     // done with panel i
		if (!issued) {                                      // (somebody is more than a panel behind)
			while (peek(B + tslot) < need_done) __builtin_amdgcn_s_sleep(1);
			issue(pa + i + D, tslot);
		}
		// publish the pieces issued during the PREVIOUS panel: they have had a panel to land
		if (pend_slot >= 0) {
			if (had_issue) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
			else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			bump(pend_slot);
		}
		pend_slot = had_issue ? tslot : -1;
		// next panel: its slot, the slot its look-ahead goes to (= the slot of this panel's predecessor's successor)
		need_done = need;                                   // panel i done: the count that `need` was for panel i
		tslot = slot + D + 1; if (tslot >= B) tslot -= B; if (tslot >= B) tslot -= B;
		slot++; if (slot == B) { slot = 0; need += 16u; }
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] == 123.456) sink[0] = acc0[1];
}

template <int B, int H>
static void run_ring(const double *Y, double *sink, int imbalance)
{
	const int kt = 2, nblocks = 16, nsplit = 8;
	const int64_t nrow = 999936, ld = 1000000, npanels = nrow / H;
	const int64_t pps = (npanels + nsplit - 1) / nsplit;
	const int nwg = nblocks * kt * nsplit;
	const size_t ldsb = (size_t) 64 * (B * H + 1) * 8 + 64;
	CHECK(hipFuncSetAttribute((const void *) combo_ring<B, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((combo_ring<B, H>), dim3(nwg), dim3(1024), ldsb, 0, Y, ld, nblocks, kt, npanels, pps, sink, imbalance);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best) best = ms;
	}
	printf("ring  %d slots x %3d rows (%zu B of LDS), counters, imbalance %d: %.3f ms (%.0f cycles per 128 rows)\n", B, H, ldsb, imbalance, best,
	       best * 1e-3 * 2.4e9 / pps * 128.0 / H);
	CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

// ------------------------------------------------------------------------------------------- E
// trace: the barrier + work loop of combo_16 (no DMA), one trip per asm block, with s_memtime stamps of
// workgroup 0 for four panels in mid-run: arrival at the barrier, release, end of every trip.  What the
// ~1200 cycles that one barrier costs (the `period` runs) are made of.
#define FR4(M, Y, Y1, A, A1_) I1(M) F1(Y, Y1) R1(A, A1_)
#define FMAI_B_READ_A ION(28) F1(108, 109) R1(12, 13) FR4(29, 110, 111, 14, 15) FR4(30, 112, 113, 16, 17) FR4(31, 114, 115, 18, 19) FR4(32, 116, 117, 20, 21) FR4(33, 118, 119, 22, 23) FR4(34, 120, 121, 24, 25) FR4(35, 122, 123, 26, 27) IOFF
#define FMAI_A_READ_B ION(28) F1(12, 13) R1(108, 109) FR4(29, 14, 15, 110, 111) FR4(30, 16, 17, 112, 113) FR4(31, 18, 19, 114, 115) FR4(32, 20, 21, 116, 117) FR4(33, 22, 23, 118, 119) FR4(34, 24, 25, 120, 121) FR4(35, 26, 27, 122, 123) IOFF
template <int VAR>
__global__ void __launch_bounds__(16 * 64)
combo_trace(int64_t npanels, double *sink, unsigned long long *trace, int first, int prio)
{
	extern __shared__ double lds[];
	constexpr int RS = 129, BUF = 64 * RS;
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	for (int i = tid; i < 2 * BUF; i += 16 * 64) lds[i] = 1.0;
	__syncthreads();
	d16 acc0 = 0.0, acc1 = 0.0;
	u32x16 ya = 0, yb = 0;
	u32x8 meta;
	const int rows[8] = {3, 17, 40, 66, 71, 90, 101, 120};
	const int cols[8] = {0, 5, 9, 12, 3, 7, 14, 15};
#pragma unroll
	for (int q = 0; q < 8; q++) meta[q] = ((uint32_t) (rows[q] * 8) << 16) | (uint32_t) (2 * cols[q]);
	const double one = 1.0000001;
	const bool rec = blockIdx.x == 0 && lane == 0;
	if (prio == 1) { if ((w >> 2) == 0) __builtin_amdgcn_s_setprio(0); else if ((w >> 2) == 1) __builtin_amdgcn_s_setprio(1); else if ((w >> 2) == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
	for (int64_t i = 0; i < npanels; i++) {
		const int buf = (int) (i & 1);
		const bool tr = rec && i >= first && i < first + 4;
		unsigned long long *t = trace + ((i - first) * 16 + w) * 8;
		if (tr) t[0] = __builtin_readcyclecounter();
		__builtin_amdgcn_s_barrier();
		if (tr) t[1] = __builtin_readcyclecounter();
		const int n = (int) ((i + 1) * 16 / 5 - i * 16 / 5);       // 3.2 trips of 16 records per panel
		const uint32_t lanebase = (uint32_t) lane * 1032u + (uint32_t) buf * (BUF * 8u);
		for (int k = 0; k < n; k++) {
			uint32_t one_trip = 1;
#define TRIP_ASM(BODY) asm volatile(LOOP_HEAD BODY LOOP_TAIL \
				     : "+{v[44:75]}"(acc0), "+{v[76:107]}"(acc1), [n] "+s"(one_trip), "+{v[12:27]}"(ya), \
				       "+{v[108:123]}"(yb) \
				     : [lb] "v"(lanebase), [one] "s"(one), "{s[28:35]}"(meta) : "memory", "scc", "s36")
			if constexpr (VAR == 1) TRIP_ASM(ADDR_A FMAI_B ADDR_B FMAI_A);
			else if constexpr (VAR == 2) TRIP_ASM(ADDR_A READ_A WAIT ADDR_B READ_B WAIT);
			else if constexpr (VAR == 4) TRIP_ASM(ADDR_A FMAI_B_READ_A WAIT ADDR_B FMAI_A_READ_B WAIT);
			else {
				if constexpr (VAR == 3) {
					switch ((((w >> 2) + k) & 3)) {
					case 0: __builtin_amdgcn_s_setprio(0); break;
					case 1: __builtin_amdgcn_s_setprio(1); break;
					case 2: __builtin_amdgcn_s_setprio(2); break;
					default: __builtin_amdgcn_s_setprio(3); break;
					}
				}
				if constexpr (VAR == 6) {          // priority = trips still to do: whoever lags goes first
					switch (n - k) {
					case 1: __builtin_amdgcn_s_setprio(0); break;
					case 2: __builtin_amdgcn_s_setprio(1); break;
					case 3: __builtin_amdgcn_s_setprio(2); break;
					default: __builtin_amdgcn_s_setprio(3); break;
					}
				}
				if constexpr (VAR == 7) {          // the same by half trips (batches of 8)
					uint32_t one_trip = 1;
					const int rem = 2 * (n - k);
					if (rem >= 6) __builtin_amdgcn_s_setprio(3); else if (rem >= 4) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
					TRIP_ASM(ADDR_A READ_A FMAI_B WAIT);
					if (rem == 2) __builtin_amdgcn_s_setprio(0);
					one_trip = 1;
					TRIP_ASM(ADDR_B READ_B FMAI_A WAIT);
					if (tr) t[2 + k] = __builtin_readcyclecounter();
					continue;
				}
				if constexpr (VAR == 5) if (k == 0) for (int z = 0; z < (w >> 2); z++) __builtin_amdgcn_s_sleep(2);
				TRIP_ASM(ADDR_A READ_A FMAI_B WAIT ADDR_B READ_B FMAI_A WAIT);
			}
			if (tr) t[2 + k] = __builtin_readcyclecounter();
		}
	}
	if (acc0[0] + acc1[3] + (double) ya[0] + (double) yb[1] == 123.456) sink[0] = acc0[1];
}

template <int VAR>
static void run_trace(double *sink, int prio, const char *what)
{
	const int64_t npanels = 977;
	const size_t ldsb = (size_t) 2 * 64 * 129 * 8;
	CHECK(hipFuncSetAttribute((const void *) combo_trace<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
	unsigned long long *trace, h[4 * 16 * 8];
	CHECK(hipMalloc(&trace, sizeof h));
	CHECK(hipMemset(trace, 0, sizeof h));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float ms = 0;
	for (int rep = 0; rep < 3; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(combo_trace<VAR>, dim3(256), dim3(1024), ldsb, 0, npanels, sink, trace, 400, prio);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		CHECK(hipEventElapsedTime(&ms, e0, e1));
	}
	CHECK(hipMemcpy(h, trace, sizeof h, hipMemcpyDeviceToHost));
	printf("trace %s, prio %d: %.3f ms for %lld panels; stamps of workgroup 0 (counter ticks relative to the first release of the panel)\n", what, prio, ms, (long long) npanels);
	for (int p = 1; p < 2; p++) {
		unsigned long long t0 = ~0ull;
		for (int w = 0; w < 16; w++) if (h[(p * 16 + w) * 8 + 1] < t0) t0 = h[(p * 16 + w) * 8 + 1];
		printf(" panel %d\n", p);
		for (int w = 0; w < 16; w++) {
			const unsigned long long *t = h + (p * 16 + w) * 8;
			printf("  wave %2d (SIMD %d, rank %d): arrive %6lld release %5lld trips", w, w & 3, w >> 2, (long long) (t[0] - t0), (long long) (t[1] - t0));
			for (int k = 0; k < 4; k++) if (t[2 + k]) printf(" %6lld", (long long) (t[2 + k] - t0));
			printf("\n");
		}
	}
	CHECK(hipFree(trace));
}

template <int MODE>
static void run_work(double *sink, int threads, const char *what)
{
	const int iters = 20000;
	const int ldsd = MODE == 4 ? 64 * 258 : 64 * 129;
	const size_t ldsb = (size_t) ldsd * 8;
	CHECK(hipFuncSetAttribute((const void *) work_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(work_kernel<MODE>, dim3(256), dim3(threads), ldsb, 0, sink, iters, ldsd);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best) best = ms;
	}
	// (record x 64 dense columns) units per CU: waves * iters * 16 -- modes 0-3, 5-8: 16 records of 64 dense columns per
	// trip of the loop; MODE 4: 8 records of 128 dense columns (two units each) per trip, 16 units as well.
	// (Round 2 counted 32 for MODE 4 and so reported half its time: "0.39-0.41 ms" was 0.78-0.82 ms -- 8 ds_read_b128 of
	// 1 KiB per trip in ~39 cycles = 210 B/clk/CU of the LDS's 256.)
	const double waves = threads / 64.0;
	const double wrec = waves * iters * 16.0;
	const double cyc = best * 1e-3 * 2.4e9;
	printf("work  %-44s %2d waves/CU: %.3f ms, %.2f cycles per (record x 64 dense columns) per CU at 2.4 GHz -> config 2a (2e8 of them over 256 CUs) %.3f ms\n",
	       what, (int) waves, best, cyc / wrec, cyc / wrec * 2e8 / 256 / 2.4e9 * 1e3);
	CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char **argv)
{
	const char *which = argc > 1 ? argv[1] : "all";
	double *sink;
	CHECK(hipMalloc(&sink, 64));
	if (!strcmp(which, "all") || !strcmp(which, "work")) {
		for (int threads : {1024, 512}) {
			run_work<0>(sink, threads, "add + ds_read_b64 + idx + fma (full)");
			run_work<1>(sink, threads, "add + ds_read_b64 + fma, fixed acc (no idx)");
			run_work<2>(sink, threads, "add + idx + fma (no LDS read)");
			run_work<3>(sink, threads, "idx + fma");
			run_work<5>(sink, threads, "fma only");
			run_work<4>(sink, threads, "2 cols/lane: add + ds_read_b128 + idx + 2 fma");
			run_work<6>(sink, threads, "plain add + ds_read_b64 + fma, fixed acc");
			run_work<7>(sink, threads, "plain add + fma, fixed acc");
			run_work<8>(sink, threads, "SDWA add + fma, fixed acc");
		}
	}
	if (!strcmp(which, "all") || !strcmp(which, "flag")) {
		const int64_t maxrow = 1048576 + 1024, K = 128;
		double *Y;
		CHECK(hipMalloc(&Y, (size_t) maxrow * K * 8 + 4096));
		CHECK(hipMemset(Y, 0, (size_t) maxrow * K * 8 + 4096));
		for (int imb = 0; imb < 2; imb++) {
			run_flag(Y, sink, imb, 1, "s_barrier per panel (trip by trip)");
			run_flag(Y, sink, imb, 0, "landed / done counters, slack 1 panel");
		}
		CHECK(hipFree(Y));
	}
	if (!strcmp(which, "all") || !strcmp(which, "combo")) {
		const int64_t maxrow = 1048576 + 1024, K = 128;
		double *Y;
		CHECK(hipMalloc(&Y, (size_t) maxrow * K * 8 + 4096));
		CHECK(hipMemset(Y, 0, (size_t) maxrow * K * 8 + 4096));
		for (int m = 0; m < 3; m++) {
			const int dma = m != 1, work = m != 2;
			run_combo(Y, sink, combo_16, 16, 16, 8, 819.2, dma, work, "16 x 40 cols, 2 y sets");
			run_combo(Y, sink, combo_16y1, 16, 16, 8, 819.2, dma, work, "16 x 40 cols, 1 y set");
			run_combo(Y, sink, combo_12, 12, 12, 10, 1075.2, dma, work, "12 x 70 cols, 1 y set");
			run_combo(Y, sink, combo_8, 8, 12, 10, 1085.4, dma, work, "8 x 106 cols, 2 y sets");
			run_combo(Y, sink, combo_8y1, 8, 12, 10, 1085.4, dma, work, "8 x 106 cols, 1 y set");
			run_combo(Y, sink, combo_8, 8, 11, 11, 1167.4, dma, work, "8 x 114 cols, 2 y sets");
			run_combo(Y, sink, combo_8y1, 8, 11, 11, 1167.4, dma, work, "8 x 114 cols, 1 y set");
		}
		CHECK(hipFree(Y));
	}
	if (!strcmp(which, "ring")) {
		const int64_t maxrow = 1048576 + 1024, K = 128;
		double *Y;
		CHECK(hipMalloc(&Y, (size_t) maxrow * K * 8 + 4096));
		CHECK(hipMemset(Y, 0, (size_t) maxrow * K * 8 + 4096));
		for (int imb = 0; imb < 2; imb++) {
			run_flag(Y, sink, imb, 1, "s_barrier per panel (trip by trip)");
			run_ring<3, 104>(Y, sink, imb);
			run_ring<4, 72>(Y, sink, imb);
			run_ring<4, 64>(Y, sink, imb);
			run_ring<6, 48>(Y, sink, imb);
		}
		CHECK(hipFree(Y));
	}
	if (!strcmp(which, "trace")) {
		run_trace<0>(sink, 0, "full");
		run_trace<0>(sink, 1, "full, priority = rank");
		run_trace<1>(sink, 0, "no LDS reads");
		run_trace<2>(sink, 0, "no FMAs");
		run_trace<3>(sink, 0, "priority rotates per trip");
		run_trace<4>(sink, 0, "reads interleaved with the FMAs");
		run_trace<5>(sink, 0, "skew rank x 128 cycles");
		run_trace<6>(sink, 0, "priority = trips still to do");
		run_trace<7>(sink, 0, "priority by half trips still to do");
	}
	if (!strcmp(which, "period")) {
		// how the cost of the barrier depends on its period: no DMA, equal record counts
		double *Y;
		CHECK(hipMalloc(&Y, 4096));
		for (int mult : {1, 2, 4, 8, 32})
			run_combo(Y, sink, combo_16, 16, 16, 8, 819.2, 0, 1, "barrier period (panels' worth)", mult);
		for (int mult : {1, 2, 4, 8, 32})
			run_combo(Y, sink, combo_16r, 16, 16, 8, 819.2, 0, 1, "adds before the wait; period", mult);
		CHECK(hipFree(Y));
		const int64_t maxrow = 1048576 + 1024, K = 128;
		CHECK(hipMalloc(&Y, (size_t) maxrow * K * 8 + 4096));
		CHECK(hipMemset(Y, 0, (size_t) maxrow * K * 8 + 4096));
		for (int rep = 0; rep < 2; rep++) {
			run_combo(Y, sink, combo_16, 16, 16, 8, 819.2, 1, 1, "with the DMA: 16 x 40 cols, 2 y sets");
			run_combo(Y, sink, combo_16r, 16, 16, 8, 819.2, 1, 1, "with the DMA: adds before the wait");
			run_combo(Y, sink, combo_16x10, 16, 16, 8, 819.2 * 0.8, 1, 1, "with the DMA: 10 records per batch");
			run_combo(Y, sink, combo_16old, 16, 16, 8, 819.2, 1, 1, "with the DMA issued by the 8 oldest wavefronts");
		}
		CHECK(hipFree(Y));
	}
	if (!strcmp(which, "all") || !strcmp(which, "stage2")) {
		const int64_t maxrow = 1048576 + 1024, K = 128;
		double *Y;
		CHECK(hipMalloc(&Y, (size_t) maxrow * K * 8 + 4096));
		CHECK(hipMemset(Y, 0, (size_t) maxrow * K * 8 + 4096));
		for (int wpb : {16, 8})
			for (int mode = 0; mode < 3; mode++)
				for (int rot : {0, 1, 5})
					run_stage2(Y, sink, wpb, rot, mode);
		CHECK(hipFree(Y));
	}
	if (!strcmp(which, "all") || !strcmp(which, "stage")) {
		const int64_t maxrow = 1048576 + 1024, K = 128;
		double *Y;
		CHECK(hipMalloc(&Y, (size_t) maxrow * K * 8 + 4096));
		CHECK(hipMemset(Y, 0, (size_t) maxrow * K * 8 + 4096));
		for (int wpb : {16, 8}) {
			run_stage(Y, sink, wpb, 0, 999936, 1000000, "col-major ld 1e6");
			run_stage(Y, sink, wpb, 0, 1048576, 1048576, "col-major ld 2^20");
			run_stage(Y, sink, wpb, 1, 999936, 0, "panel-major [p][k][128]");
		}
		for (int skew : {1, 2, 4, 16})
			run_stage(Y, sink, 16, 0, 999936, 1000000, "col-major ld 1e6", skew, 0, 0);
		run_stage(Y, sink, 16, 0, 999936, 1000000, "col-major ld 1e6", 0, 1, 0);
		run_stage(Y, sink, 16, 0, 999936, 1000000, "col-major ld 1e6", 1, 1, 0);
		for (int aux : {1, 2, 3}) {
			run_stage(Y, sink, 16, 0, 999936, 1000000, "col-major ld 1e6", 0, 0, aux);
			run_stage(Y, sink, 16, 0, 999936, 1000000, "col-major ld 1e6", 1, 0, aux);
		}
	}
	return 0;
}
