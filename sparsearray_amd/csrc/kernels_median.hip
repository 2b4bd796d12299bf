// colMedians of an SVT_SparseMatrix on the CSC device layout.
//
// Reference: pure R, one leaf at a time (.colMedians_SVT_SparseMatrix /
// .padded_median / .positive_padded_median, R/SparseArray-matrixStats.R:690-784;
// its own TODO asks for a C version behind C_colStats_SVT).  .padded_median(x,
// padding) is "median(c(x, integer(padding)))" without realising the zeros: the
// n = length(x) + padding values are order statistics of [negatives | zeros |
// positives], the median is the middle one or the mean of the two middle ones.
// NA rule (:714-719): na.rm drops NA/NaN from the nonzeros (the padding keeps its
// size); otherwise any NA/NaN gives NA_real_.  n == 0 gives NA_real_ (:721-722).
//
// Device: the nonzero values of the columns that need it are copied as f64 keys with every
// NA/NaN turned into one canonical positive NaN (sorts last), sorted per column by rocprim's
// segmented radix sort, and one thread per column picks the order statistics by three
// binary searches (first key >= 0, first key > 0, first NaN).
// Most columns of a sparse matrix never get that far: when the middle ranks fall among the
// zeros (fewer than half of the column's values positive, fewer than half negative) the
// median is 0, which a counting pass over the values decides (median_count_kernel); only
// the other columns keep a non-empty segment for the sort.
// Roofline: HBM; algorithmic bytes = 8 per nonzero for the count pass, 8 + 8 for the key
// copy, and 16 per sort pass for the columns that need one.
#include "svt_common.h"

#include <string.h>
#include <rocprim/rocprim.hpp>

// Keys of the columns that need a sort (a wavefront per column; the others -- at BASELINE config 2
// all of them -- return at once: copying every value cost 0.3 ms of a 0.55 ms colMedians there).
template <typename T>
__global__ void __launch_bounds__(256)
median_key_kernel(const int64_t *__restrict__ col_ptr, const T *__restrict__ val, int64_t ncol,
		  const int64_t *__restrict__ seg_b, const int64_t *__restrict__ seg_e,
		  double *__restrict__ keys)
{
	const int lane = threadIdx.x & 63;
	const int64_t j = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
	if (j >= ncol || seg_e[j] == seg_b[j]) return;
	const int64_t beg = col_ptr[j], end = col_ptr[j + 1];
	for (int64_t k = beg + lane; k < end; k += 64) {
		double d;
		if (sizeof(T) == 8) {
			d = (double) val[k];
		} else {
			const int v = (int) val[k];
			d = v == NA_INT ? NAN : (double) v;
		}
		if (d != d) d = __longlong_as_double(0x7FF8000000000000LL);
		keys[k] = d;
	}
}

// One wavefront per column: negatives, positives, NA/NaN among the stored values.  Writes the
// result where no order statistic of the nonzeros is needed (NA rule, empty column, both middle
// ranks among the zeros) and gives every other column its sort segment [seg_b, seg_e).
template <typename T>
__global__ void __launch_bounds__(256)
median_count_kernel(const int64_t *__restrict__ col_ptr, const T *__restrict__ val, int64_t nrow,
		    int64_t ncol, int na_rm, double *__restrict__ out,
		    int64_t *__restrict__ seg_b, int64_t *__restrict__ seg_e)
{
	const int lane = threadIdx.x & 63;
	const int64_t j = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
	if (j >= ncol) return;
	const int64_t beg = col_ptr[j], end = col_ptr[j + 1];
	long long neg = 0, pos = 0, nan = 0;
	// (four loads per lane in flight: one wavefront per column with a single load each kept 16 KB per CU
	// on the way, 3.1 TB/s; colMedians at BASELINE config 2 is this pass alone -- every median is a zero)
	for (int64_t k0 = beg; k0 < end; k0 += 256) {
		T raw[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int64_t k = k0 + u * 64 + lane;
			raw[u] = k < end ? val[k] : (T) 0;
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			if (k0 + u * 64 + lane >= end) continue;
			double d;
			if (sizeof(T) == 8) d = (double) raw[u];
			else { const int v = (int) raw[u]; d = v == NA_INT ? NAN : (double) v; }
			if (d != d) nan++;
			else if (d < 0.0) neg++;
			else if (d > 0.0) pos++;
		}
	}
	neg = wave_sum_ll(neg); pos = wave_sum_ll(pos); nan = wave_sum_ll(nan);
	neg = __shfl(neg, 0, 64); pos = __shfl(pos, 0, 64); nan = __shfl(nan, 0, 64);
	if (lane != 0) return;
	const int64_t len = end - beg, v = len - nan, padding = nrow - len, n = v + padding;
	int64_t b = beg, e = beg;                        // empty segment: nothing to sort
	if ((!na_rm && nan > 0) || n == 0) {
		out[j] = svt_na_real();
	} else {
		const int64_t z = v - neg - pos + padding;   // stored + implicit zeros
		const int64_t lo = (n - 1) >> 1, hi = n >> 1;
		if (lo >= neg && hi < neg + z) out[j] = 0.0;
		else { out[j] = -1.0; e = end; }             // decided by median_pick_kernel
	}
	seg_b[j] = b; seg_e[j] = e;
}

__global__ void median_pick_kernel(const int64_t *__restrict__ col_ptr, const double *__restrict__ keys,
				   int64_t nrow, int64_t ncol, int na_rm, double *__restrict__ out,
				   const int64_t *__restrict__ seg_e)
{
	const int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= ncol) return;
	const int64_t beg = col_ptr[j], end = col_ptr[j + 1];
	if (seg_e[j] != end || end == beg) return;       // decided by the counting pass
	const double *__restrict__ s = keys + beg;
	const int64_t len = end - beg;
	// first NaN, first key >= 0, first key > 0
	int64_t lo = 0, hi = len;
	while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (s[m] == s[m]) lo = m + 1; else hi = m; }
	const int64_t v = lo;                            // valid (non-NA) stored values
	if ((!na_rm && v < len)) { out[j] = svt_na_real(); return; }
	const int64_t padding = nrow - len;
	const int64_t n = v + padding;
	if (n == 0) { out[j] = svt_na_real(); return; }
	lo = 0; hi = v;
	while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (s[m] < 0.0) lo = m + 1; else hi = m; }
	const int64_t a = lo;                            // negatives
	hi = v;
	while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (s[m] <= 0.0) lo = m + 1; else hi = m; }
	const int64_t z0 = lo - a, z = z0 + padding;     // stored zeros (not expected), all zeros
	auto elem = [&](int64_t r) -> double {
		if (r < a) return s[r];
		if (r < a + z) return 0.0;
		return s[a + z0 + (r - a - z)];
	};
	if (n & 1) out[j] = elem((n - 1) >> 1);
	else out[j] = (elem((n >> 1) - 1) + elem(n >> 1)) * 0.5;     // (:707 mean of the two, :757)
}

size_t colmedians_ws_bytes(int64_t nnz, int64_t ncol)
{
	size_t tmp = 0;
	const int64_t n = nnz > 0 ? nnz : 1;
	(void) rocprim::segmented_radix_sort_keys(NULL, tmp, (const double *) NULL, (double *) NULL,
						  (unsigned int) n, (unsigned int) (ncol > 0 ? ncol : 1),
						  (const int64_t *) NULL, (const int64_t *) NULL);
	return (size_t) n * 16 + tmp + (size_t) (ncol > 0 ? ncol : 1) * 16 + 1024;
}

int launch_colmedians(const int64_t *col_ptr, const void *val, int Rtype, int64_t nrow, int64_t ncol,
		      int64_t nnz, int na_rm, double *out, void *ws, hipStream_t s)
{
	if (ncol <= 0)
		return 0;
	if (nnz > 0x7FFFFFFFLL || ncol > 0x7FFFFFFFLL)
		return svt_set_error("colMedians: more than 2^31-1 nonzeros or columns");
	double *k_in = (double *) ws;
	double *k_out = k_in + (nnz > 0 ? nnz : 1);
	int64_t *seg_b = (int64_t *) (((uintptr_t) (k_out + (nnz > 0 ? nnz : 1)) + 255) & ~(uintptr_t) 255);
	int64_t *seg_e = seg_b + ncol;
	void *tmp = (void *) (((uintptr_t) (seg_e + ncol) + 255) & ~(uintptr_t) 255);
	{
		const unsigned nbc = (unsigned) ((ncol + 3) / 4);
		if (Rtype == SVT_REALSXP)
			hipLaunchKernelGGL(median_count_kernel<double>, dim3(nbc), dim3(256), 0, s, col_ptr,
					   (const double *) val, nrow, ncol, na_rm, out, seg_b, seg_e);
		else
			hipLaunchKernelGGL(median_count_kernel<int>, dim3(nbc), dim3(256), 0, s, col_ptr,
					   (const int *) val, nrow, ncol, na_rm, out, seg_b, seg_e);
	}
	if (nnz > 0) {
		const unsigned nb = (unsigned) ((ncol + 3) / 4);
		if (Rtype == SVT_REALSXP)
			hipLaunchKernelGGL(median_key_kernel<double>, dim3(nb), dim3(256), 0, s, col_ptr, (const double *) val,
					   ncol, seg_b, seg_e, k_in);
		else
			hipLaunchKernelGGL(median_key_kernel<int>, dim3(nb), dim3(256), 0, s, col_ptr, (const int *) val,
					   ncol, seg_b, seg_e, k_in);
		size_t tmp_bytes = 0;
		HIP_TRY(rocprim::segmented_radix_sort_keys(NULL, tmp_bytes, k_in, k_out, (unsigned int) nnz, (unsigned int) ncol,
							   seg_b, seg_e));
		HIP_TRY(rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, k_in, k_out, (unsigned int) nnz, (unsigned int) ncol,
							   seg_b, seg_e, 0u, 64u, s));
	}
	hipLaunchKernelGGL(median_pick_kernel, dim3((unsigned) ((ncol + 255) / 256)), dim3(256), 0, s,
			   col_ptr, k_out, nrow, ncol, na_rm, out, seg_e);
	HIP_TRY(hipGetLastError());
	return 0;
}
