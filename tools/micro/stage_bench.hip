// Micro-benchmark: how fast can every CU stream row panels of a column-major
// dense matrix into LDS with LDS-DMA (global_load_lds_dwordx4), double-buffered,
// one barrier per panel -- the staging half of the crossprod kernel without
// the record loop -- and do the three candidate LDS images come out right?
//   variant 0: k-row stride 130 doubles (16-byte aligned rows)
//   variant 1: same, rows of k-rows 16..31 / 48..63 shifted by one (source - 8 B)
//   variant 2: k-row stride 129 doubles (odd k-rows start 8 mod 16)
// hipcc -O3 --offload-arch=gfx950 -o stage_bench stage_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int VARIANT, int NPIECE, int DEPTH>
__global__ void __launch_bounds__(1024)
stage_kernel(const double *__restrict__ Y, int64_t ldY, int64_t nrow, int nblocks, int kt,
	     int64_t panels_per_split, int RP, int loops, int check, unsigned long long *bad_out,
	     double *sink)
{
	extern __shared__ double lds[];
	constexpr int RS = VARIANT == 2 ? 129 : 130;
	constexpr int BUF = 16 * NPIECE * RS;        // doubles per buffer
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int L = blockIdx.x;
	const int xcd = L % 8, j = L / 8;
	const int b = j % nblocks, u = j / nblocks;
	const int kh = u % kt, sp = (u / kt) * 8 + xcd;
	(void) b;
	const int64_t npanels = (nrow + RP - 1) / RP;
	const int64_t pa = (int64_t) sp * panels_per_split;
	int64_t pb = pa + panels_per_split;
	if (pb > npanels - 1) pb = npanels - 1;      // interior panels only
	if (pa >= pb) return;
	const int k0 = kh * 64;
	unsigned long long bad = 0;
	double acc = 0.0;

	auto issue = [&](int64_t p, int buf) {
#pragma unroll
		for (int q = 0; q < NPIECE; q++) {
			const int kk = w * NPIECE + q;
			const int shift = (VARIANT == 1 && ((kk >> 4) & 1)) ? 1 : 0;
			const double *src = Y + (int64_t) (k0 + kk) * ldY + p * RP - shift + lane * 2;
			double *dst = lds + buf * BUF + kk * RS;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
							 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
		}
	};
	constexpr int NB = DEPTH + 1;
	for (int it = 0; it < loops; it++) {
	for (int d = 0; d < DEPTH; d++)
		if (pa + d < pb) issue(pa + d, d);
	for (int64_t p = pa; p < pb; p++) {
		const int buf = (int) ((p - pa) % NB);
		// panels p .. p+DEPTH-1 are in flight; wait for the oldest
		if (p + DEPTH - 1 < pb) {
			if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPIECE) : "memory");
			else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NPIECE) : "memory");
		} else {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		__builtin_amdgcn_s_barrier();
		if (p + DEPTH < pb) issue(p + DEPTH, (int) ((p - pa + DEPTH) % NB));
		if (check) {
			// every thread verifies 4 elements of the landed panel
			for (int e = 0; e < 4; e++) {
				const int idx = (tid * 4 + e) * 7 % (16 * NPIECE * 127);
				const int kk = idx / 127, r = idx % 127;
				const int shift = (VARIANT == 1 && ((kk >> 4) & 1)) ? 1 : 0;
				const double got = lds[buf * BUF + kk * RS + r + shift];
				const double want = Y[(int64_t) (k0 + kk) * ldY + p * RP + r];
				if (got != want) bad++;
			}
		} else {
			acc += lds[buf * BUF + (lane % (16 * NPIECE)) * RS + (w & 7)];
		}
	}
	__builtin_amdgcn_s_barrier();
	}
	if (check) { if (bad) atomicAdd(bad_out, bad); }
	else if (acc == 123.456) sink[0] = acc;
}

int main(int argc, char **argv)
{
	const int64_t nrow = argc > 3 ? atoll(argv[3]) : 1000000, K = 128;
	const int loops = argc > 4 ? atoi(argv[4]) : 1;
	const int nblocks = argc > 1 ? atoi(argv[1]) : 16;
	const int splits_per_xcd = argc > 2 ? atoi(argv[2]) : 1;
	const int RP = 127, kt = 2;
	double *Y, *sink;
	unsigned long long *bad;
	CHECK(hipMalloc(&Y, (size_t) nrow * K * 8 + 4096));
	CHECK(hipMalloc(&sink, 64));
	CHECK(hipMalloc(&bad, 8));
	std::vector<double> h((size_t) nrow * K);
	for (size_t i = 0; i < h.size(); i++) h[i] = (double) (i % 1000003) + 0.25;
	CHECK(hipMemcpy(Y, h.data(), h.size() * 8, hipMemcpyHostToDevice));
	const int nsplit = 8 * splits_per_xcd;
	const int64_t npanels = (nrow + RP - 1) / RP;
	const int64_t pps = (npanels + nsplit - 1) / nsplit;
	const int nwg = nblocks * kt * nsplit;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	struct Cfg { int variant, npiece, depth; void (*kern)(const double *, int64_t, int64_t, int, int, int64_t, int, int, int, unsigned long long *, double *); };
	const Cfg cfgs[] = {
		{2, 4, 1, stage_kernel<2, 4, 1>}, {2, 2, 1, stage_kernel<2, 2, 1>}, {2, 1, 1, stage_kernel<2, 1, 1>},
		{2, 2, 2, stage_kernel<2, 2, 2>}, {2, 2, 3, stage_kernel<2, 2, 3>}, {2, 1, 3, stage_kernel<2, 1, 3>},
		{2, 1, 6, stage_kernel<2, 1, 6>},
	};
	for (const Cfg &c : cfgs) {
		const size_t ldsb = (size_t) (c.depth + 1) * 16 * c.npiece * 129 * 8;
		if (ldsb > 160 * 1024) { printf("skip\n"); continue; }
		CHECK(hipFuncSetAttribute((const void *) c.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
		CHECK(hipMemset(bad, 0, 8));
		hipLaunchKernelGGL(c.kern, dim3(nwg), dim3(1024), ldsb, 0, Y, nrow, nrow, nblocks, kt, pps, RP, 1, 1, bad, sink);
		CHECK(hipDeviceSynchronize());
		unsigned long long hb = 0;
		CHECK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
		float best = 1e30f;
		for (int rep = 0; rep < 5; rep++) {
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(c.kern, dim3(nwg), dim3(1024), ldsb, 0, Y, nrow, nrow, nblocks, kt, pps, RP, loops, 0, bad, sink);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		const double bytes = (double) loops * nblocks * nrow * K * 8 * c.npiece / 4;
		const double panels = (double) pps * loops;
		printf("npiece %d (%d KB/panel) depth %d: nwg %d: mismatches %llu, %.3f ms, %.1f GB staged, %.1f TB/s, %.1f B/clk/CU, %.0f cycles/panel\n",
		       c.npiece, 16 * c.npiece, c.depth, nwg, hb, best, bytes / 1e9, bytes / best / 1e9,
		       bytes / 256 / (best * 1e-3 * 2.4e9), best * 1e-3 * 2.4e9 / panels);
	}
	return 0;
}
