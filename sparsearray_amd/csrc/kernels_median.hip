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
// Device: the nonzero values are copied as f64 keys with every NA/NaN turned into
// one canonical positive NaN (sorts last), sorted per column by hipcub's segmented
// radix sort, and one thread per column picks the order statistics by three
// binary searches (first key >= 0, first key > 0, first NaN).
// Roofline: HBM; algorithmic bytes = 8 per nonzero read + 8 written per sort pass.
#include "svt_common.h"
#include <hipcub/hipcub.hpp>

template <typename T>
__global__ void median_key_kernel(const T *__restrict__ val, int64_t nnz, double *__restrict__ keys)
{
	const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nnz) return;
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

__global__ void median_pick_kernel(const int64_t *__restrict__ col_ptr, const double *__restrict__ keys,
				   int64_t nrow, int64_t ncol, int na_rm, double *__restrict__ out)
{
	const int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= ncol) return;
	const int64_t beg = col_ptr[j], end = col_ptr[j + 1];
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
	(void) hipcub::DeviceSegmentedRadixSort::SortKeys(NULL, tmp, (const double *) NULL, (double *) NULL,
							  (int) n, (int) (ncol > 0 ? ncol : 1),
							  (const int64_t *) NULL, (const int64_t *) NULL);
	return (size_t) n * 16 + tmp + 512;
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
	void *tmp = (void *) (((uintptr_t) (k_out + (nnz > 0 ? nnz : 1)) + 255) & ~(uintptr_t) 255);
	if (nnz > 0) {
		const unsigned nb = (unsigned) ((nnz + 255) / 256);
		if (Rtype == SVT_REALSXP)
			hipLaunchKernelGGL(median_key_kernel<double>, dim3(nb), dim3(256), 0, s, (const double *) val, nnz, k_in);
		else
			hipLaunchKernelGGL(median_key_kernel<int>, dim3(nb), dim3(256), 0, s, (const int *) val, nnz, k_in);
		size_t tmp_bytes = 0;
		HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(NULL, tmp_bytes, k_in, k_out, (int) nnz, (int) ncol,
								   col_ptr, col_ptr + 1));
		HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(tmp, tmp_bytes, k_in, k_out, (int) nnz, (int) ncol,
								   col_ptr, col_ptr + 1, 0, 64, s));
	}
	hipLaunchKernelGGL(median_pick_kernel, dim3((unsigned) ((ncol + 255) / 256)), dim3(256), 0, s,
			   col_ptr, k_out, nrow, ncol, na_rm, out);
	HIP_TRY(hipGetLastError());
	return 0;
}
