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
// Device (round 5: no library sort): a counting pass (one wavefront per column) finds the
// negatives, positives and NA/NaN among the stored values.  Most columns of a sparse matrix
// are decided there: when the middle ranks fall among the zeros (fewer than half of the
// column's values positive, fewer than half negative) the median is 0.  For the other
// columns one workgroup per column SELECTS the one or two order statistics it needs from
// the column where it lies -- a most-significant-digit-first radix select over the
// order-preserving 64-bit image of the doubles (digits of 11, 11, 11, 11, 10, 10 bits:
// counting passes over the column with a histogram in LDS until the bin of the wanted rank
// holds at most 1024 keys, which are then collected into LDS and ranked there; the second
// middle value, when n is even, comes from the same candidates or is the smallest key above
// their bin).  Nothing is copied and nothing is sorted (rounds 2-4: a key copy + rocprim's
// segmented radix sort of 64-bit keys, eight read + write passes over the values).
// Roofline: HBM; algorithmic bytes = 8 per nonzero for the count pass and typically 3 x 8
// (at most 8 x 8) per nonzero of an undecided column (short columns stay in the L2).
#include "svt_common.h"

#include <string.h>

// One wavefront per column: negatives, positives, NA/NaN among the stored values.  Writes the
// result where no order statistic of the nonzeros is needed (NA rule, empty column, both middle
// ranks among the zeros); every other column gets todo[j] = 1 and its counts (cnt_neg, cnt_valid):
// decided by median_select_kernel.
template <typename T>
__global__ void __launch_bounds__(256)
median_count_kernel(const int64_t *__restrict__ col_ptr, const T *__restrict__ val, int64_t nrow,
		    int64_t ncol, int na_rm, double *__restrict__ out,
		    int64_t *__restrict__ cnt_neg, int64_t *__restrict__ cnt_pos, int64_t *__restrict__ cnt_nan,
		    int *__restrict__ todo)
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
	int undecided = 0;
	if ((!na_rm && nan > 0) || n == 0) {
		out[j] = svt_na_real();
	} else {
		const int64_t z = v - neg - pos + padding;   // stored + implicit zeros
		const int64_t lo = (n - 1) >> 1, hi = n >> 1;
		if (lo >= neg && hi < neg + z) out[j] = 0.0;
		else undecided = 1;                          // decided by median_select_kernel
	}
	cnt_neg[j] = neg; cnt_pos[j] = pos; cnt_nan[j] = nan; todo[j] = undecided;
}

// ---- radix select -----------------------------------------------------------------------------------
#define MSEL_NT 256
#define MSEL_BINS 2048

template <typename T>
__device__ inline bool msel_key(T raw, unsigned long long *key)
{
	double d;
	if (sizeof(T) == 8) d = (double) raw;
	else { const int v = (int) raw; d = v == NA_INT ? NAN : (double) v; }
	if (d != d || d == 0.0)
		return false;                            // NA / NaN, and a stored zero (it counts among the zeros)
	*key = f64_to_ordered(d);
	return true;
}

// Block-wide exclusive prefix of one count per thread (MSEL_NT threads); returns the thread's prefix, *total
// = the sum.  wsum: 4 words of LDS.
__device__ inline unsigned msel_block_scan(unsigned x, unsigned *wsum, unsigned *total)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	unsigned incl = x;
	for (int o = 1; o < 64; o <<= 1) {
		const unsigned t = __shfl_up(incl, o, 64);
		if (lane >= o) incl += t;
	}
	if (lane == 63) wsum[w] = incl;
	__syncthreads();
	unsigned before = 0, all = 0;
	for (int i = 0; i < MSEL_NT / 64; i++) {
		const unsigned t = wsum[i];
		if (i < w) before += t;
		all += t;
	}
	__syncthreads();
	*total = all;
	return before + incl - x;
}

#define MSEL_CAND 1024        // candidates finished in LDS

// The keys of ranks k and k + 1 (0-based, ascending) among the NONZERO, non-NA values of val[beg, end) -- *key1 only when
// want_next (the caller knows that rank k + 1 exists).  Counting passes, most significant digit first (11, 11, 11, 11,
// 10, 10 bits), until the bin that holds rank k has at most MSEL_CAND keys; those are then collected into LDS in one more
// pass over the column (which also finds the smallest key ABOVE the bin, for rank k + 1 when k is the bin's last) and
// ranked there.  Doubles of one sign and exponent differ in their mantissa: two counting passes (22 bits) leave
// n / 1024 candidates of a column of n values -- three passes over the column instead of seven.
// Called by all MSEL_NT threads of the workgroup with the same arguments.  Returns false when the keys never got few
// enough (more than MSEL_CAND equal keys): *key0 is exact after the six passes, *key1 is then left to the caller.
template <typename T>
__device__ bool msel_select(const T *__restrict__ val, int64_t beg, int64_t end, unsigned k, bool want_next,
			    unsigned *hist, unsigned *wsum, unsigned *found, unsigned long long *cand,
			    unsigned long long *key0, unsigned long long *key1)
{
	unsigned long long prefix = 0;
	int shift = 64;
	for (int pass = 0; pass < 6; pass++) {
		const int w = pass < 4 ? 11 : 10;
		const int hi_shift = shift;              // bits [hi_shift, 64) of the key are fixed by `prefix`
		shift -= w;
		const unsigned nb = 1u << w, mask = nb - 1;
		for (unsigned i = threadIdx.x; i < nb; i += MSEL_NT) hist[i] = 0;
		__syncthreads();
		// (four loads per thread in flight: a long column is walked by ONE workgroup)
		for (int64_t i0 = beg; i0 < end; i0 += 4 * MSEL_NT) {
			T raw[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int64_t i = i0 + u * MSEL_NT + threadIdx.x;
				raw[u] = i < end ? val[i] : (T) 0;       // (a zero is skipped by msel_key)
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				unsigned long long key;
				if (!msel_key<T>(raw[u], &key))
					continue;
				if (pass == 0 || (key >> hi_shift) == (prefix >> hi_shift))
					atomicAdd(&hist[(unsigned) (key >> shift) & mask], 1u);
			}
		}
		__syncthreads();
		// the digit whose bin holds rank k: a thread owns nb / MSEL_NT consecutive bins
		const unsigned per = nb / MSEL_NT, b0 = threadIdx.x * per;
		unsigned mine = 0;
		for (unsigned i = 0; i < per; i++) mine += hist[b0 + i];
		unsigned total;
		const unsigned before = msel_block_scan(mine, wsum, &total);
		if (k >= before && k < before + mine) {
			unsigned run = before;
			for (unsigned i = 0; i < per; i++) {
				const unsigned c = hist[b0 + i];
				if (k < run + c) { found[0] = b0 + i; found[1] = run; found[2] = c; break; }
				run += c;
			}
		}
		__syncthreads();
		prefix |= (unsigned long long) found[0] << shift;
		k -= found[1];
		const unsigned ncand = found[2];
		__syncthreads();
		if (ncand <= MSEL_CAND && shift > 0) {
			// collect the bin's keys (bits [shift, 64) equal to the prefix) and the smallest key above the bin
			if (threadIdx.x == 0) found[3] = 0;
			__syncthreads();
			unsigned long long above = ~0ull;
			for (int64_t i0 = beg; i0 < end; i0 += 4 * MSEL_NT) {
				T raw[4];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int64_t i = i0 + u * MSEL_NT + threadIdx.x;
					raw[u] = i < end ? val[i] : (T) 0;
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					unsigned long long key;
					if (!msel_key<T>(raw[u], &key))
						continue;
					const unsigned long long hi = key >> shift, want = prefix >> shift;
					if (hi == want) cand[atomicAdd(&found[3], 1u)] = key;
					else if (hi > want && key < above) above = key;
				}
			}
			if (want_next) {
				const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
				for (int o = 32; o > 0; o >>= 1) {
					const unsigned long long t = __shfl_down(above, o, 64);
					above = t < above ? t : above;
				}
				// (hist is free again: its first words carry the wavefronts' minima)
				if (lane == 0) ((unsigned long long *) hist)[wv] = above;
			}
			__syncthreads();
			if (want_next) {
				above = ~0ull;
				for (int i = 0; i < MSEL_NT / 64; i++) {
					const unsigned long long t = ((unsigned long long *) hist)[i];
					above = t < above ? t : above;
				}
			}
			// rank the candidates: cand[i] is the key of local rank k iff (keys below it) <= k < (keys below or equal)
			if (threadIdx.x == 0) { found[0] = 0; found[1] = 0; }
			__syncthreads();
			unsigned long long *res = (unsigned long long *) (hist + 64);       // [0] key of rank k, [1] of rank k + 1
			for (unsigned i = threadIdx.x; i < ncand; i += MSEL_NT) {
				const unsigned long long mk = cand[i];
				unsigned lt = 0, le = 0;
				for (unsigned jx = 0; jx < ncand; jx++) {
					const unsigned long long o = cand[jx];
					lt += o < mk; le += o <= mk;
				}
				if (lt <= k && k < le) res[0] = mk;                     // (equal keys write the same value)
				if (lt <= k + 1 && k + 1 < le) { res[1] = mk; found[1] = 1; }
			}
			__syncthreads();
			*key0 = res[0];
			*key1 = found[1] ? res[1] : above;      // rank k + 1 past the bin's last key: the smallest key above the bin
			__syncthreads();
			return true;
		}
	}
	*key0 = prefix;
	*key1 = 0;
	return false;
}

// One workgroup per undecided column (grid-stride over the columns): the virtual sorted column is
// [negatives | z zeros | positives]; rank r < neg is the r-th smallest stored value, rank r >= neg + z the
// (r - z)-th smallest NONZERO stored value.
template <typename T>
__global__ void __launch_bounds__(MSEL_NT)
median_select_kernel(const int64_t *__restrict__ col_ptr, const T *__restrict__ val, int64_t nrow, int64_t ncol,
		     const int64_t *__restrict__ cnt_neg, const int64_t *__restrict__ cnt_pos,
		     const int64_t *__restrict__ cnt_nan, const int *__restrict__ todo, double *__restrict__ out)
{
	__shared__ __attribute__((aligned(16))) unsigned hist[MSEL_BINS];      // (also read as 64-bit words below)
	__shared__ unsigned wsum[MSEL_NT / 64];
	__shared__ unsigned found[4];
	__shared__ unsigned long long red[2 * (MSEL_NT / 64)];
	__shared__ unsigned long long cand[MSEL_CAND];
	for (int64_t j = blockIdx.x; j < ncol; j += gridDim.x) {
		if (!todo[j])
			continue;                                // (the same answer in every thread)
		const int64_t beg = col_ptr[j], end = col_ptr[j + 1];
		const int64_t neg = cnt_neg[j], pos = cnt_pos[j];
		const int64_t nz = neg + pos;                    // nonzero, non-NA stored values
		// (columns holding NA / NaN come here only under na.rm: those entries are dropped, the padding keeps its size)
		const int64_t len = end - beg, nan = cnt_nan[j];
		const int64_t zeros = (nrow - len) + (len - nan - nz);   // implicit zeros + stored zeros
		const int64_t n = nz + zeros;
		const int64_t lo = (n - 1) >> 1, hi = n >> 1;
		// rank in the virtual column -> rank among the nonzero stored values, or -1 for "a zero"
		const int64_t klo = lo < neg ? lo : lo < neg + zeros ? -1 : lo - zeros;
		const int64_t khi = hi < neg ? hi : hi < neg + zeros ? -1 : hi - zeros;
		double vlo = 0.0, vhi = 0.0;
		unsigned long long key_lo = 0, key_next = 0;
		bool have_next = false;
		if (klo >= 0) {
			have_next = msel_select<T>(val, beg, end, (unsigned) klo, khi == klo + 1, hist, wsum, found, cand, &key_lo, &key_next);
			vlo = ordered_to_f64(key_lo);
		}
		if (khi < 0) {
			vhi = 0.0;
		} else if (khi == klo) {
			vhi = vlo;
		} else if (klo >= 0 && have_next) {
			vhi = ordered_to_f64(key_next);          // (both ranks from the same candidates)
		} else if (klo >= 0) {
			// khi == klo + 1: the same key again if ranks <= klo + 1 are all covered by keys <= key_lo, else the
			// smallest key above it.  One pass: count of keys <= key_lo, minimum of the keys > key_lo.
			unsigned long long cnt = 0, nxt = ~0ull;
			for (int64_t i0 = beg; i0 < end; i0 += 4 * MSEL_NT) {
				T raw[4];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int64_t i = i0 + u * MSEL_NT + threadIdx.x;
					raw[u] = i < end ? val[i] : (T) 0;
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					unsigned long long key;
					if (!msel_key<T>(raw[u], &key))
						continue;
					if (key <= key_lo) cnt++;
					else if (key < nxt) nxt = key;
				}
			}
			const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
			for (int o = 32; o > 0; o >>= 1) {
				cnt += __shfl_down(cnt, o, 64);
				const unsigned long long t = __shfl_down(nxt, o, 64);
				nxt = t < nxt ? t : nxt;
			}
			if (lane == 0) { red[w] = cnt; red[MSEL_NT / 64 + w] = nxt; }
			__syncthreads();
			cnt = 0; nxt = ~0ull;
			for (int i = 0; i < MSEL_NT / 64; i++) {
				cnt += red[i];
				nxt = red[MSEL_NT / 64 + i] < nxt ? red[MSEL_NT / 64 + i] : nxt;
			}
			__syncthreads();
			vhi = (int64_t) cnt > khi ? vlo : ordered_to_f64(nxt);
		} else {
			unsigned long long key_hi = 0, unused = 0;
			(void) msel_select<T>(val, beg, end, (unsigned) khi, false, hist, wsum, found, cand, &key_hi, &unused);
			vhi = ordered_to_f64(key_hi);
		}
		if (threadIdx.x == 0)
			out[j] = (n & 1) ? vlo : (vlo + vhi) * 0.5;          // (:707 mean of the two, :757)
	}
}

size_t colmedians_ws_bytes(int64_t nnz, int64_t ncol)
{
	(void) nnz;
	// [negatives per column][positives per column][NA / NaN per column][undecided flag per column]
	return (size_t) (ncol > 0 ? ncol : 1) * 28 + 1024;
}

int launch_colmedians(const int64_t *col_ptr, const void *val, int Rtype, int64_t nrow, int64_t ncol,
		      int64_t nnz, int na_rm, double *out, void *ws, hipStream_t s)
{
	if (ncol <= 0)
		return 0;
	if (nnz > 0x7FFFFFFFLL || ncol > 0x7FFFFFFFLL)
		return svt_set_unsupported("colMedians: more than 2^31-1 nonzeros or columns");
	int64_t *cnt_neg = (int64_t *) (((uintptr_t) ws + 255) & ~(uintptr_t) 255);
	int64_t *cnt_pos = cnt_neg + ncol, *cnt_nan = cnt_pos + ncol;
	int *todo = (int *) (cnt_nan + ncol);
	const unsigned nbc = (unsigned) ((ncol + 3) / 4);
	// the undecided columns, one workgroup each, a few rounds of them in flight
	const unsigned nbs = (unsigned) (ncol < 4096 ? ncol : 4096);
	if (Rtype == SVT_REALSXP) {
		hipLaunchKernelGGL(median_count_kernel<double>, dim3(nbc), dim3(256), 0, s, col_ptr,
				   (const double *) val, nrow, ncol, na_rm, out, cnt_neg, cnt_pos, cnt_nan, todo);
		if (nnz > 0)
			hipLaunchKernelGGL(median_select_kernel<double>, dim3(nbs), dim3(MSEL_NT), 0, s, col_ptr,
					   (const double *) val, nrow, ncol, cnt_neg, cnt_pos, cnt_nan, todo, out);
	} else {
		hipLaunchKernelGGL(median_count_kernel<int>, dim3(nbc), dim3(256), 0, s, col_ptr,
				   (const int *) val, nrow, ncol, na_rm, out, cnt_neg, cnt_pos, cnt_nan, todo);
		if (nnz > 0)
			hipLaunchKernelGGL(median_select_kernel<int>, dim3(nbs), dim3(MSEL_NT), 0, s, col_ptr,
					   (const int *) val, nrow, ncol, cnt_neg, cnt_pos, cnt_nan, todo, out);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}
