// Micro-benchmark: cost of a wave-uniform dynamically indexed accumulator update
//   mode 0: d16 vector indexing (compiler: s_set_gpr_idx_on + v_mov)
//   mode 1: switch over 16 static registers (compiler: scalar branch tree)
//   mode 2: no indexing (single accumulator) -- floor
//   mode 3: hand-written gpr-index mode: one mode-on region per batch of 4,
//           v_fma_f64 with relative src2/dst on a pinned register block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(const unsigned *__restrict__ idx, int n, double *out)
{
	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned *__restrict__ my = idx + (size_t) (blockIdx.x * 16 + w) * n;
	d16 acc = 0.0;
	double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0, a9 = 0, a10 = 0,
	       a11 = 0, a12 = 0, a13 = 0, a14 = 0, a15 = 0;
	double y = 1.0 + lane;
	for (int i = 0; i < n; i += 4) {
		unsigned c4[4];
#pragma unroll
		for (int q = 0; q < 4; q++) c4[q] = my[i + q];
		if (MODE == 3) {
			const unsigned i0 = (c4[0] & 15) * 2, i1 = (c4[1] & 15) * 2, i2 = (c4[2] & 15) * 2, i3 = (c4[3] & 15) * 2;
			const double v0 = (double) (c4[0] >> 4), v1 = (double) (c4[1] >> 4),
				     v2 = (double) (c4[2] >> 4), v3 = (double) (c4[3] >> 4);
			asm volatile("s_set_gpr_idx_on %[i0], gpr_idx(SRC2,DST)\n\t"
				     "v_fma_f64 v[64:65], %[v0], %[y], v[64:65]\n\t"
				     "s_set_gpr_idx_idx %[i1]\n\t"
				     "v_fma_f64 v[64:65], %[v1], %[y], v[64:65]\n\t"
				     "s_set_gpr_idx_idx %[i2]\n\t"
				     "v_fma_f64 v[64:65], %[v2], %[y], v[64:65]\n\t"
				     "s_set_gpr_idx_idx %[i3]\n\t"
				     "v_fma_f64 v[64:65], %[v3], %[y], v[64:65]\n\t"
				     "s_set_gpr_idx_off"
				     : "+{v[64:95]}"(acc)
				     : [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3),
				       [v0] "s"(v0), [v1] "s"(v1), [v2] "s"(v2), [v3] "s"(v3), [y] "v"(y)
				     : "m0");
		}
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const unsigned x = c4[q];
			const int c = x & 15;
			const double v = (double) (x >> 4);
			if (MODE == 3) {
				continue;
			} else if (MODE == 0) {
				acc[c] = __builtin_fma(v, y, acc[c]);
			} else if (MODE == 1) {
				switch (c) {
#define C(N) case N: a##N = __builtin_fma(v, y, a##N); break;
				C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14)
				default: a15 = __builtin_fma(v, y, a15);
				}
			} else {
				a0 = __builtin_fma(v, y, a0);
			}
		}
	}
	double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;
	for (int j = 0; j < 16; j++) s += acc[j];
	out[(size_t) blockIdx.x * 1024 + threadIdx.x] = s;
}

int main()
{
	const int n = 4096, nblk = 512;
	unsigned *h = (unsigned *) malloc((size_t) nblk * 16 * n * 4), *d;
	for (size_t i = 0; i < (size_t) nblk * 16 * n; i++) h[i] = (unsigned) rand();
	double *out;
	hipMalloc(&d, (size_t) nblk * 16 * n * 4);
	hipMalloc(&out, (size_t) nblk * 1024 * 8);
	hipMemcpy(d, h, (size_t) nblk * 16 * n * 4, hipMemcpyHostToDevice);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int mode = 0; mode < 4; mode++) {
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
			if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(1024), 0, 0, d, n, out);
			if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(1024), 0, 0, d, n, out);
			if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(1024), 0, 0, d, n, out);
			if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nblk), dim3(1024), 0, 0, d, n, out);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			// updates per wave: n ; waves: nblk*16 ; SIMDs: 1024 ; 2 rounds of 256 blocks
			double upd = (double) n * nblk * 16;
			if (rep == 1) { double *ho = (double *) malloc((size_t) nblk * 1024 * 8); hipMemcpy(ho, out, (size_t) nblk * 1024 * 8, hipMemcpyDeviceToHost); double cs = 0; for (size_t t = 0; t < (size_t) nblk * 1024; t++) cs += ho[t]; printf("   checksum %.6e\n", cs); free(ho); }
			printf("mode %d rep %d: %.3f ms  -> %.1f SIMD-cycles per update (at 2.1 GHz, 1024 SIMDs)\n",
			       mode, rep, ms, ms * 1e-3 * 2.1e9 * 1024 / upd);
		}
	}
	return 0;
}
