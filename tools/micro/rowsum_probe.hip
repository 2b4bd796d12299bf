// rowsum(x, group) at BASELINE config 3's shape (1e6 x 1e4 @ 1 %, 1000 groups): is the gather of the group id
// (one 64-byte sector from L2 per nonzero, what bounds the product's kernel) avoidable by letting the
// wavefronts of a workgroup walk DIFFERENT columns through the SAME window of rows, so that the ids they look
// up sit in the CU's vector L1?
//
//   A  the library's form: one 256-thread workgroup per column, accumulators in LDS
//   B  one workgroup = C columns, one wavefront each, all of them inside the same window of W rows; a barrier
//      per window; the next chunk of (row, value) is loaded a window ahead
//
// hipcc -O3 --offload-arch=gfx950 -o rowsum_probe rowsum_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// synthetic operand: column j holds PER nonzeros, the i-th in row i * STEP + (hash % STEP): ascending, ~uniform
__global__ void make_csc(int64_t ncol, int per, int step, int32_t *row_idx, double *val)
{
	const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= ncol * per) return;
	const int64_t j = k / per, i = k % per;
	row_idx[k] = (int32_t) (i * step + mix32((uint32_t) (j * 1000003 + i)) % step);
	val[k] = 1.0 + (double) (mix32((uint32_t) k) & 1023) * (1.0 / 1024);
}
// the same with the gaps of a uniformly random column (geometric, mean STEP): thread per column
__global__ void make_csc_random(int64_t ncol, int per, int step, int64_t nrow, int32_t *row_idx)
{
	const int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= ncol) return;
	int64_t r = -1;
	const double l1p = log(1.0 - 1.0 / step);
	for (int i = 0; i < per; i++) {
		const double u = ((double) mix32((uint32_t) (j * 1000003 + i) ^ 0x9e3779b9u) + 0.5) / 4294967296.0;
		r += 1 + (int64_t) (log(u) / l1p);
		if (r > nrow - (per - i)) r = nrow - (per - i);
		row_idx[j * per + i] = (int32_t) r;
	}
}
__global__ void make_groups(int64_t nrow, int ngroup, uint16_t *g16)
{
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (r < nrow) g16[r] = (uint16_t) (mix32((uint32_t) r * 2654435761u) % ngroup);
}

__global__ void __launch_bounds__(256)
rowsum_A(const int32_t *__restrict__ row_idx, const double *__restrict__ val, int per, int ngroup,
	 const uint16_t *__restrict__ g16, double *__restrict__ out)
{
	extern __shared__ double acc[];
	const int64_t j = blockIdx.x;
	for (int g = threadIdx.x; g < ngroup; g += blockDim.x) acc[g] = 0.0;
	__syncthreads();
	const int64_t beg = j * per, end = beg + per;
	for (int64_t k = beg + threadIdx.x; k < end; k += blockDim.x) atomicAdd(&acc[g16[row_idx[k]]], val[k]);
	__syncthreads();
	for (int g = threadIdx.x; g < ngroup; g += blockDim.x) out[j * ngroup + g] = acc[g];
}

template <int C>
__global__ void __launch_bounds__(C * 64)
rowsum_B(const int32_t *__restrict__ row_idx, const double *__restrict__ val, int per, int ngroup, int64_t ncol,
	 int64_t nrow, int W, const uint16_t *__restrict__ g16, double *__restrict__ out, int alias_rows)
{
	extern __shared__ double acc[];                 // [C][ngroup]
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t j = (int64_t) blockIdx.x * C + w;
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) acc[g] = 0.0;
	__syncthreads();
	double *mine = acc + w * ngroup;
	const bool have = j < ncol;
	const int64_t jr = alias_rows ? (int64_t) blockIdx.x * C : j;          // (probe: every wavefront looks up the same rows)
	const int64_t beg = have ? j * per : 0, end = have ? beg + per : 0;
	const int64_t rshift = (jr - j) * per;
	int64_t k = beg;
	int32_t r = k + lane < end ? row_idx[k + lane + rshift] : 0x7FFFFFFF;
	double v = k + lane < end ? val[k + lane] : 0.0;
	for (int64_t R = W; ; R += W) {
		for (;;) {
			const bool in = r < R;
			const int cnt = __popcll(__ballot(in));
			// the next chunk goes out before this one's lookups
			const int64_t kn = k + cnt;
			const int32_t rn = kn + lane < end ? row_idx[kn + lane + rshift] : 0x7FFFFFFF;
			const double vn = kn + lane < end ? val[kn + lane] : 0.0;
			if (in) atomicAdd(&mine[g16[r]], v);
			k = kn; r = rn; v = vn;
			if (cnt < 64) break;
		}
		if (R >= nrow) break;
		__syncthreads();
	}
	__syncthreads();
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) {
		const int64_t jj = (int64_t) blockIdx.x * C + g / ngroup;
		if (jj < ncol) out[jj * ngroup + g % ngroup] = acc[g];
	}
}

// D: B with the window's group ids staged in LDS (16-bit, W <= 16384 rows = 32 KB) and looked up there
template <int C, int W>
__global__ void __launch_bounds__(C * 64)
rowsum_D(const int32_t *__restrict__ row_idx, const double *__restrict__ val, int per, int ngroup, int64_t ncol,
	 int64_t nrow, const uint16_t *__restrict__ g16, double *__restrict__ out)
{
	extern __shared__ double acc[];                 // [C][ngroup] then W ids
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t j = (int64_t) blockIdx.x * C + w;
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) acc[g] = 0.0;
	uint16_t *ids = (uint16_t *) (acc + C * ngroup);
	double *mine = acc + w * ngroup;
	const bool have = j < ncol;
	const int64_t beg = have ? j * per : 0, end = have ? beg + per : 0;
	int64_t k = beg;
	int32_t r = k + lane < end ? row_idx[k + lane] : 0x7FFFFFFF;
	double v = k + lane < end ? val[k + lane] : 0.0;
	for (int64_t R0 = 0; R0 < nrow; R0 += W) {
		const int64_t R = R0 + W;
		__syncthreads();                                // everybody done with the previous window's ids
		for (int x = threadIdx.x * 8; x < W; x += C * 64 * 8)
			if (R0 + x + 8 <= nrow) *(uint4 *) (ids + x) = *(const uint4 *) (g16 + R0 + x);
			else for (int q = 0; q < 8; q++) if (R0 + x + q < nrow) ids[x + q] = g16[R0 + x + q];
		__syncthreads();
		for (;;) {
			const bool in = r < R;
			const int cnt = __popcll(__ballot(in));
			const int64_t kn = k + cnt;
			const int32_t rn = kn + lane < end ? row_idx[kn + lane] : 0x7FFFFFFF;
			const double vn = kn + lane < end ? val[kn + lane] : 0.0;
			if (in) atomicAdd(&mine[ids[r - R0]], v);
			k = kn; r = rn; v = vn;
			if (cnt < 64) break;
		}
	}
	__syncthreads();
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) {
		const int64_t jj = (int64_t) blockIdx.x * C + g / ngroup;
		if (jj < ncol) out[jj * ngroup + g % ngroup] = acc[g];
	}
}

// B2: B with TWO chunks of (row, value) loaded ahead.  A chunk is 64 consecutive nonzeros from the cursor; the cursor
// moves by the number of rows that fell inside the window, so the chunk after next is loaded from cursor + 64
// (an upper bound of the next cursor) and realigned with a wavefront shuffle when the window cut the chunk short.
template <int C>
__global__ void __launch_bounds__(C * 64)
rowsum_B2(const int32_t *__restrict__ row_idx, const double *__restrict__ val, int per, int ngroup, int64_t ncol,
	  int64_t nrow, int W, const uint16_t *__restrict__ g16, double *__restrict__ out, int alias_rows)
{
	extern __shared__ double acc[];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t j = (int64_t) blockIdx.x * C + w;
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) acc[g] = 0.0;
	__syncthreads();
	double *mine = acc + w * ngroup;
	const bool have = j < ncol;
	const int64_t beg = have ? j * per : 0, end = have ? beg + per : 0;
	// whole chunks only: the window test masks lanes, the cursor always moves by 64; a chunk that straddles a
	// window's end is finished in the next window (its remaining lanes kept in the registers)
	int64_t k = beg;
	int32_t r0 = k + lane < end ? row_idx[k + lane] : 0x7FFFFFFF;
	double v0 = k + lane < end ? val[k + lane] : 0.0;
	int32_t r1 = k + 64 + lane < end ? row_idx[k + 64 + lane] : 0x7FFFFFFF;
	double v1 = k + 64 + lane < end ? val[k + 64 + lane] : 0.0;
	bool todo = true;                                  // this lane's element of chunk 0 not yet added
	for (int64_t R = W; ; R += W) {
		for (;;) {
			const bool in = todo && r0 < R;
			if (in) { atomicAdd(&mine[g16[r0]], v0); todo = false; }
			// chunk 0 finished (every lane added or past the end of the column)?
			const bool left = todo && r0 != 0x7FFFFFFF;
			if (__ballot(left)) break;                 // the rest of it belongs to a later window
			if (k >= end) break;
			k += 64;
			r0 = r1; v0 = v1; todo = true;
			r1 = k + 64 + lane < end ? row_idx[k + 64 + lane] : 0x7FFFFFFF;
			v1 = k + 64 + lane < end ? val[k + 64 + lane] : 0.0;
		}
		if (R >= nrow) break;
		__syncthreads();
	}
	__syncthreads();
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) {
		const int64_t jj = (int64_t) blockIdx.x * C + g / ngroup;
		if (jj < ncol) out[jj * ngroup + g % ngroup] = acc[g];
	}
}

int main(int argc, char **argv)
{
	const int64_t nrow = 1100000, ncol = 10000;       // (random gaps: a column may end a few per cent past 1e6)
	const int per = 10000, step = 100, ngroup = 1000;
	const int64_t nnz = ncol * per;
	int32_t *row_idx; double *val, *outA, *outB; uint16_t *g16;
	CHECK(hipMalloc(&row_idx, nnz * 4)); CHECK(hipMalloc(&val, nnz * 8));
	CHECK(hipMalloc(&outA, ncol * ngroup * 8)); CHECK(hipMalloc(&outB, ncol * ngroup * 8));
	CHECK(hipMalloc(&g16, nrow * 2));
	hipLaunchKernelGGL(make_csc, dim3((unsigned) ((nnz + 255) / 256)), dim3(256), 0, 0, ncol, per, step, row_idx, val);
	hipLaunchKernelGGL(make_groups, dim3((unsigned) ((nrow + 255) / 256)), dim3(256), 0, 0, nrow, ngroup, g16);
	if (argc > 1) hipLaunchKernelGGL(make_csc_random, dim3((unsigned) ((ncol + 63) / 64)), dim3(64), 0, 0, ncol, per, step, nrow, row_idx);
	CHECK(hipDeviceSynchronize());
	printf("rows: %s\n", argc > 1 ? "geometric gaps (a uniformly random column)" : "one per stripe of 100 rows");
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	auto timeit = [&](auto launch, const char *what) {
		float best = 1e30f;
		for (int rep = 0; rep < 5; rep++) {
			CHECK(hipEventRecord(e0));
			launch();
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (rep > 0 && ms < best) best = ms;
		}
		CHECK(hipGetLastError());
		printf("%-70s %.3f ms\n", what, best);
	};
	timeit([&] { hipLaunchKernelGGL(rowsum_A, dim3((unsigned) ncol), dim3(256), ngroup * 8, 0, row_idx, val, per, ngroup, g16, outA); },
	       "A  workgroup per column (the library's kernel)");
	std::vector<double> hA(ncol * ngroup), hB(ncol * ngroup);
	CHECK(hipMemcpy(hA.data(), outA, hA.size() * 8, hipMemcpyDeviceToHost));
	auto runB = [&](auto kern, int C, int W, int alias) {
		const size_t lds = (size_t) C * ngroup * 8;
		CHECK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		char what[128];
		snprintf(what, sizeof what, "B  %2d columns per workgroup, windows of %5d rows%s", C, W, alias ? ", all wavefronts on the same rows (probe)" : "");
		CHECK(hipMemset(outB, 0, hB.size() * 8));
		timeit([&] { hipLaunchKernelGGL(kern, dim3((unsigned) ((ncol + C - 1) / C)), dim3(C * 64), lds, 0, row_idx, val, per, ngroup, ncol, nrow, W, g16, outB, alias); }, what);
		if (!alias) {
			CHECK(hipMemcpy(hB.data(), outB, hB.size() * 8, hipMemcpyDeviceToHost));
			double worst = 0;
			for (size_t i = 0; i < hA.size(); i++) { const double d = fabs(hA[i] - hB[i]); if (d > worst) worst = d; }
			printf("      largest difference from A: %.3g\n", worst);
		}
	};
	for (int W : {12800, 25600, 51200, 102400, 204800}) {
		runB(rowsum_B<16>, 16, W, 0);
		runB(rowsum_B<14>, 14, W, 0);
	}
	runB(rowsum_B<16>, 16, 102400, 1);
	auto runD = [&](auto kern, int C, int W) {
		const size_t lds = (size_t) C * ngroup * 8 + (size_t) W * 2;
		CHECK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		char what[128];
		snprintf(what, sizeof what, "D  %2d columns per workgroup, ids of %5d rows in LDS", C, W);
		CHECK(hipMemset(outB, 0, hB.size() * 8));
		timeit([&] { hipLaunchKernelGGL(kern, dim3((unsigned) ((ncol + C - 1) / C)), dim3(C * 64), lds, 0, row_idx, val, per, ngroup, ncol, nrow, g16, outB); }, what);
		CHECK(hipMemcpy(hB.data(), outB, hB.size() * 8, hipMemcpyDeviceToHost));
		double worst = 0;
		for (size_t i = 0; i < hA.size(); i++) { const double d = fabs(hA[i] - hB[i]); if (d > worst) worst = d; }
		printf("      largest difference from A: %.3g\n", worst);
	};
	runD(rowsum_D<14, 16384>, 14, 16384);
	runD(rowsum_D<14, 8192>, 14, 8192);
	runD(rowsum_D<16, 16384>, 16, 16384);
	runD(rowsum_D<15, 16384>, 15, 16384);
	runD(rowsum_D<14, 24576>, 14, 24576);

	return 0;
}
