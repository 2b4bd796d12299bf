// A %*% B for two sparse operands in the CSC device layout, B much sparser than a dense matrix
// (svt %*% svt2 of BASELINE config 3: A 1e6 x 1e4 @ 1 %, B 1e4 x 128 @ 1 % = 12800 nonzeros).
//
// The reference computes x %*% y as .crossprod2_SparseMatrix_SparseMatrix(t(x), y) ->
// C_crossprod2_SVT_SVT (src/SparseMatrix_mult.c:1037-1101): it expands the leaves of one operand into a
// dense buffer and walks the leaves of the other over it (crossprod2_Lpp_* / crossprod2_Rpp_*, :728-820):
// per output cell the sum, in ascending order of the inner index, of the products over the nonzeros of the
// walked leaf -- products with the buffer's zeros add exact zeros.  With finite operands that is the sum
// over the inner indices where BOTH operands hold a nonzero, and that is what this kernel adds up, without
// a dense operand:  out[:, k] = sum over the nonzeros (j, b) of B[:, k] of  b * A[:, j].
// (The dense route -- densify B, product on the panel-blocked layout of t(A) -- multiplies every nonzero of A
// with all K columns: 1.28e10 multiply-adds and a 2.2 ms transposition + a 2 ms layout build where 1.3e8
// multiply-adds and one pass over the offsets do.)
//
// Workgroup = a panel of P rows x KW columns of the result, kept in LDS (KW * P * 8 bytes <= 64 KB: 8192 rows of
// one column for tall operands, two workgroups per CU).  The part of
// column j of A that falls into the panel is one contiguous run of its (ascending) offsets; the run bounds
// come from the table of launch_rowpanel_table() (kernels_rowstats.hip), one pass over A's offsets.  A group
// of G lanes takes one nonzero of B at a time -- the pairs (j, b) of the workgroup's KW columns, dealt round
// the groups -- and adds b * A[run] into the column's LDS image (ds_add_f64: two pairs can meet in a row);
// the image leaves as whole, coalesced columns.  Not a sum in the reference's order: the additions of one
// cell come in the order the lane groups get to them (differences of the last bits between runs; integer
// operands give the reference's result exactly as long as the sums stay below 2^53, where the order of the
// additions does not matter).
// Config 3 (tools/debug/spmm_sweep.sh, whole call): 8192 x 1 with 32 lanes per run 0.86 ms; 8192 x 2 (128 KB) 1.05; 16384 x 1
// 0.95; 4096 x 2 1.01; 64 lanes per run 0.97; the dense route 2.35-2.45 ms + 2.2 ms t(A) + 2.0 ms layout once per A.
//
// A non-finite value or an NA ANYWHERE in either operand changes what the reference computes (its dirty-leaf
// loops multiply the implicit zeros too, src/SparseVec_dotprod.c:48-65): *flag goes up and the caller takes the
// dense route.  Who looks at which values (round 4; a separate scan of all of A was 0.15 of a 0.85 ms call):
//   * the product kernel at every value it reads anyway -- all of B, and the leaves of A that some column of B
//     refers to;
//   * the pass that builds the table of run bounds (it streams A's offsets) at the values of the leaves of A
//     that NO column of B refers to (28 % of them at config 3: a byte map of the referenced leaves is made from
//     B's offsets first) -- or at all of them when the table is prepared for any B (SpmmPlan).
#include "svt_common.h"

#define SPMM_NT 1024
#define SPMM_LDS (64 * 1024)          // per workgroup: two workgroups per CU
#ifndef SPMM_U
#define SPMM_U 4
#endif
#ifndef SPMM_T
#define SPMM_T 1
#endif

static size_t spmm_table_bytes(int64_t nrow, int64_t ninner)
{
	// the table of run bounds for the shortest panels this file uses (64 rows)
#ifdef SVT_TUNING
	int ps = 7;                                     // (tuning builds: SVT_SPMM_PS may shorten the panels down to 128 rows)
#else
	int ps = 12;                                    // (the shortest full panels spmm_shape() picks)
#endif
	while (ps > 6 && ((int64_t) 1 << ps) >= 2 * (nrow > 0 ? nrow : 1)) ps--;
	const int64_t npan = (nrow + ((int64_t) 1 << ps) - 1) >> ps;
	return ((size_t) (ninner > 0 ? ninner : 1) * (size_t) (npan + 1) * 4 + 255) / 256 * 256;
}

// [table of run bounds][byte map of the leaves of A that B refers to]
size_t spmm_ws_bytes(int64_t nrow, int64_t ninner)
{
	return spmm_table_bytes(nrow, ninner) + (size_t) (ninner > 0 ? ninner : 1) + 256;
}

// ref[j] = 1 for every leaf j of A that a nonzero of B sits in row j of
__global__ void __launch_bounds__(256)
spmm_ref_kernel(const int32_t *__restrict__ b_idx, int64_t b_nnz, uint8_t *__restrict__ ref)
{
	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < b_nnz; i += (int64_t) gridDim.x * blockDim.x)
		ref[b_idx[i]] = 1;
}

// the same for a small B, zeroing the map first: ONE workgroup, one launch (a memset in front of the kernel above is a
// blit kernel with a bubble before it: ~10 us of a 0.73 ms call)
__global__ void __launch_bounds__(1024)
spmm_ref_small_kernel(const int32_t *__restrict__ b_idx, int64_t b_nnz, uint8_t *__restrict__ ref, int64_t ninner,
		      int *__restrict__ zero2)
{
	if (zero2 != NULL && threadIdx.x < 2) zero2[threadIdx.x] = 0;   // (the two flag words at the head of the workspace)
	for (int64_t i = threadIdx.x; i < ninner; i += blockDim.x) ref[i] = 0;
	__syncthreads();
	for (int64_t i = threadIdx.x; i < b_nnz; i += blockDim.x) ref[b_idx[i]] = 1;
}

static int g_spmm_ps = 13, g_spmm_nt = SPMM_NT, g_spmm_kw = 16;
static void spmm_knobs(void)
{
	static bool done = false;
	if (done) return;
	done = true;
	// (tuning build only.  Round 4 at config 3, prepared product: 8192 rows x 1024 threads 0.64 ms; 4096 x 512 0.72;
	// 4096 x 1024 0.78; 4096 x 256 0.90; 2048 x 512 0.88; 2048 x 256 1.03 -- shorter panels = shorter runs, a larger table)
#ifdef SVT_TUNING
	if (getenv("SVT_SPMM_PS")) g_spmm_ps = atoi(getenv("SVT_SPMM_PS"));
	if (getenv("SVT_SPMM_NT")) g_spmm_nt = atoi(getenv("SVT_SPMM_NT"));
	if (getenv("SVT_SPMM_KW")) g_spmm_kw = atoi(getenv("SVT_SPMM_KW"));
#endif
	if (g_spmm_kw < 1) g_spmm_kw = 1;
	if (g_spmm_ps < 7 || g_spmm_ps > 13) g_spmm_ps = 13;
	if (g_spmm_nt != 256 && g_spmm_nt != 512 && g_spmm_nt != 1024) g_spmm_nt = SPMM_NT;
}

template <typename T> __device__ inline bool spmm_bad(T v);
template <> __device__ inline bool spmm_bad<double>(double v) { return !(fabs(v) <= 1.7976931348623157e308); }
template <> __device__ inline bool spmm_bad<int>(int v) { return v == NA_INT; }

// *flag = 1 when a value is NaN / Inf / NA (doubles) or NA_integer_ (ints): ALL values of an operand, not only
// those the product reads -- an Inf in a leaf of x poisons its dot products with every leaf of y, matched or not
// (compute_dotprods2_with_left_double_leaf, src/SparseMatrix_mult.c:632-652: a dirty leaf goes to the loops that
// multiply the implicit zeros; a clean expanded leaf meets the other operand's Inf as 0 * Inf)
template <typename T>
__global__ void __launch_bounds__(256)
spmm_scan_values_kernel(const T *__restrict__ val, int64_t n, int *__restrict__ flag)
{
	bool bad = false;
	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
		bad |= spmm_bad<T>(val[i]);
	if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) *flag = 1;
}

static void spmm_scan_values(const void *val, int Rtype, int64_t n, int *flag, hipStream_t s)
{
	if (n <= 0) return;
	int64_t nb = (n + 256 * 8 - 1) / (256 * 8);
	if (nb > 256 * 16) nb = 256 * 16;
	if (Rtype == SVT_REALSXP)
		hipLaunchKernelGGL(spmm_scan_values_kernel<double>, dim3((unsigned) nb), dim3(256), 0, s, (const double *) val, n, flag);
	else
		hipLaunchKernelGGL(spmm_scan_values_kernel<int>, dim3((unsigned) nb), dim3(256), 0, s, (const int *) val, n, flag);
}

template <typename TA, typename TB>
__global__ void __launch_bounds__(SPMM_NT)
spmm_csc_csc_kernel(SpmmArgs a, int KW, int G, int nkb, int order)
{
	extern __shared__ double acc[];                 // [KW][P]
	const int tid = threadIdx.x;
	const int P = 1 << a.ps;
	// 1-D grid of npan x nkb workgroups (nkb = blocks of KW result columns).  order 0 (the product's): panels fastest.
	// order 1: the column blocks of a panel are neighbours in launch order -- the workgroups in flight then cover a few
	// panels of A for ALL columns of B, so that the runs of A several columns of B name (1.28 reads per nonzero of A at
	// config 3) could come from the caches: measured in round 5, no gain (0.608 against 0.599 ms prepared; tuning knob)
	const int64_t L = blockIdx.x;
	const int64_t q = order ? L / nkb : L % a.npan, kb = order ? L % nkb : L / a.npan;
	const int64_t r0 = q << a.ps;
	const int64_t k0 = kb * KW;
	const int kw = (int) (a.K - k0 < KW ? a.K - k0 : KW);
	const int np = (int) (a.nrow - r0 < P ? a.nrow - r0 : P);
	const int NT = blockDim.x;
	for (int x = tid * 2; x < kw * P; x += NT * 2) *(double2 *) (acc + x) = make_double2(0.0, 0.0);   // (P is even)
	// the pairs of the workgroup's columns, flattened: pair t belongs to column kk with pre[kk] <= t < pre[kk + 1]
	__shared__ int64_t bbeg[17];
	__shared__ int32_t pre[17];
	__shared__ int stop;
	if (tid == 0) {
		// an earlier pass (or another workgroup of this launch) has found a value that voids the result: the caller takes
		// the dense route.  ONE read per workgroup, shared through LDS: the wavefronts of a workgroup must agree, the
		// flag can rise between their reads
		stop = *(volatile const int *) a.flag;
		int32_t run = 0;
		for (int kk = 0; kk < kw; kk++) {
			bbeg[kk] = a.b_ptr[k0 + kk];
			pre[kk] = run;
			run += (int32_t) (a.b_ptr[k0 + kk + 1] - a.b_ptr[k0 + kk]);
		}
		pre[kw] = run;
	}
	__syncthreads();
	if (stop != 0)
		return;
	const int npairs = pre[kw];
	const int grp = tid / G, sl = tid % G, ngrp = NT / G;
	const TA *__restrict__ av = (const TA *) a.a_val;
	const TB *__restrict__ bv = (const TB *) a.b_val;
	const int32_t *__restrict__ pt0 = a.pt + q * a.ninner, *__restrict__ pt1 = pt0 + a.ninner;
	bool bad = false;
	// SPMM_U pairs per group in flight: their bounds are fetched together, then their runs
	for (int t0 = grp; t0 < npairs; t0 += SPMM_U * ngrp) {
		int64_t xb[SPMM_U], xe[SPMM_U];
		double bval[SPMM_U];
		int kkof[SPMM_U];
#pragma unroll
		for (int u = 0; u < SPMM_U; u++) {
			const int t = t0 + u * ngrp;
			xb[u] = xe[u] = 0; bval[u] = 0.0; kkof[u] = 0;
			if (t < npairs) {
				int kk = 0;
				while (kk + 1 < kw && pre[kk + 1] <= t) kk++;
				const int64_t pos = bbeg[kk] + (t - pre[kk]);
				const int64_t j = a.b_idx[pos];
				const TB b = bv[pos];
				const int64_t base = a.a_ptr[j];
				xb[u] = base + pt0[j] + sl; xe[u] = base + pt1[j];
				bval[u] = (double) b; kkof[u] = kk;
				bad |= spmm_bad<TB>(b);
			}
		}
		// SPMM_T trips of every pair's run fetched together.  Measured at config 3 (a run holds ~2.5 trips of 32
		// lanes): T = 1 0.60 ms (54 VGPRs, two workgroups per CU); T = 3 0.875 (82 VGPRs: one workgroup per CU);
		// T = 2 held to 64 VGPRs 0.95 (20 of them spilled) -- the round trips in a row are not what the kernel waits for.
		bool more = true;
		while (more) {
			TA v[SPMM_U][SPMM_T];
			int r[SPMM_U][SPMM_T];
#pragma unroll
			for (int u = 0; u < SPMM_U; u++)
#pragma unroll
				for (int t = 0; t < SPMM_T; t++) {
					const int64_t x = xb[u] + (int64_t) t * G;
					if (x < xe[u]) { v[u][t] = av[x]; r[u][t] = (int) (a.a_idx[x] - r0); }
				}
			more = false;
#pragma unroll
			for (int u = 0; u < SPMM_U; u++) {
#pragma unroll
				for (int t = 0; t < SPMM_T; t++)
					if (xb[u] + (int64_t) t * G < xe[u]) {
						bad |= spmm_bad<TA>(v[u][t]);
						atomicAdd(&acc[kkof[u] * P + r[u][t]], (double) v[u][t] * bval[u]);
					}
				if (xb[u] < xe[u]) {
					xb[u] += (int64_t) SPMM_T * G;
					more |= xb[u] < xe[u];
				}
			}
		}
	}
	if (__ballot(bad) != 0 && (tid & 63) == 0) *a.flag = 1;
	__syncthreads();
	for (int kk = 0; kk < kw; kk++) {
		double *__restrict__ dst = a.out + (k0 + kk) * a.ldo + r0;
		if ((((uintptr_t) dst) & 15) == 0) {
			// the result is written once and not read again by this launch: 16-byte non-temporal stores, so that the
			// 1 GB of it does not push A's runs out of the caches
			typedef double d2 __attribute__((ext_vector_type(2)));
			for (int x = tid * 2; x < np; x += NT * 2) {
				if (x + 1 < np) {
					const d2 v = *(const d2 *) (acc + kk * P + x);
					__builtin_nontemporal_store(v, (d2 *) (dst + x));
				} else
					dst[x] = acc[kk * P + x];
			}
		} else
			for (int x = tid; x < np; x += NT) dst[x] = acc[kk * P + x];
	}
}

// Shape of the launch for an operand with nrow rows (shared by the preparation and the product)
static void spmm_shape(int64_t nrow, int64_t K, int *ps_out, int64_t *npan_out, int *KW_out)
{
	spmm_knobs();
	int ps = g_spmm_ps;                             // 8192-row panels; shorter operands: one or two panels
	while (ps > 6 && ((int64_t) 1 << ps) >= 2 * nrow) ps--;
	const int64_t P = (int64_t) 1 << ps, npan = (nrow + P - 1) >> ps;
	int KW = (int) (SPMM_LDS / (P * 8));
	if (KW > 16) KW = 16;
	if (KW > g_spmm_kw) KW = g_spmm_kw;
	if (KW > K) KW = (int) K;
	if (KW < 1) KW = 1;
	// with few panels: fewer columns per workgroup, so that the grid fills the chip
	while (KW > 1 && npan * ((K + KW - 1) / KW) < 512) KW = (KW + 1) / 2;
	*ps_out = ps; *npan_out = npan; *KW_out = KW;
}

// What depends on A alone: the table of run bounds (into ws) and the scan of its values (raises *flag).
int launch_spmm_prepare(const SpmmArgs &a, int64_t a_nnz, void *ws, hipStream_t s)
{
	if (a.nrow <= 0)
		return 0;
	int ps, KW; int64_t npan;
	spmm_shape(a.nrow, 1, &ps, &npan, &KW);
	if (!launch_rowpanel_table_scan(a.a_ptr, a.a_idx, a.a_val, a.a_type, a.ninner, a_nnz, npan, ps, (int32_t *) ws,
					NULL, a.flag, s))
		spmm_scan_values(a.a_val, a.a_type, a_nnz, a.flag, s);
	HIP_TRY(hipGetLastError());
	return 0;
}

// The same for ONE product with B: only the leaves of A that B does not refer to are looked at here -- the
// product kernel sees the values of the others.
int launch_spmm_prepare_for(const SpmmArgs &a, int64_t a_nnz, int64_t b_nnz, void *ws, hipStream_t s, int *zero2)
{
	if (a.nrow <= 0) {
		if (zero2 != NULL) HIP_TRY(hipMemsetAsync(zero2, 0, 8, s));
		return 0;
	}
	int ps, KW; int64_t npan;
	spmm_shape(a.nrow, 1, &ps, &npan, &KW);
	uint8_t *ref = (uint8_t *) ws + spmm_table_bytes(a.nrow, a.ninner);
	if (a.ninner <= ((int64_t) 1 << 18) && b_nnz <= ((int64_t) 1 << 18)) {
		hipLaunchKernelGGL(spmm_ref_small_kernel, dim3(1), dim3(1024), 0, s, a.b_idx, b_nnz, ref, a.ninner, zero2);
	} else {
		if (zero2 != NULL) HIP_TRY(hipMemsetAsync(zero2, 0, 8, s));
		HIP_TRY(hipMemsetAsync(ref, 0, (size_t) (a.ninner > 0 ? a.ninner : 1), s));
		if (b_nnz > 0) {
			int64_t nb = (b_nnz + 255) / 256;
			if (nb > 1024) nb = 1024;
			hipLaunchKernelGGL(spmm_ref_kernel, dim3((unsigned) nb), dim3(256), 0, s, a.b_idx, b_nnz, ref);
		}
	}
	if (!launch_rowpanel_table_scan(a.a_ptr, a.a_idx, a.a_val, a.a_type, a.ninner, a_nnz, npan, ps, (int32_t *) ws,
					ref, a.flag, s))
		spmm_scan_values(a.a_val, a.a_type, a_nnz, a.flag, s);
	HIP_TRY(hipGetLastError());
	return 0;
}

// The product on a prepared A (ws as left by launch_spmm_prepare / _prepare_for); it looks at every value it reads
// (raises *flag).  Every cell of out[0 .. nrow) x [0 .. K) is written unless *flag was up already.
int launch_spmm_product(SpmmArgs a, int64_t a_nnz, int64_t b_nnz, const void *ws, hipStream_t s)
{
	if (a.nrow <= 0 || a.K <= 0)
		return 0;
	// the kernel counts the nonzeros of a workgroup's columns of B in 32 bits
	if (b_nnz >= (int64_t) 2147483647)
		return svt_set_unsupported("sparse x sparse product: the second operand holds 2^31 nonzeros or more");
	int ps, KW; int64_t npan;
	spmm_shape(a.nrow, a.K, &ps, &npan, &KW);
	if (npan >= (int64_t) 2147483647)
		return svt_set_unsupported("sparse x sparse product: too many row panels for one launch");
	const int64_t P = (int64_t) 1 << ps;
	a.pt = (const int32_t *) ws; a.npan = npan; a.ps = ps;
	// lanes per run: the largest power of two <= half its mean length (81 nonzeros at config 3: three trips of
	// 32 lanes = 96 slots, not two of 64 = 128)
	int G = 64;
	if (a.ninner > 0) {
		const double run = (double) a_nnz / ((double) a.ninner * (double) npan);
		while (G > 8 && run < 2.0 * G) G >>= 1;
	}
#ifdef SVT_TUNING
	if (getenv("SVT_SPMM_G")) G = atoi(getenv("SVT_SPMM_G"));
#endif
	const size_t lds = (size_t) KW * P * 8;
	const int64_t nkb = (a.K + KW - 1) / KW;
	if (npan * nkb >= (int64_t) 2147483647)
		return svt_set_unsupported("sparse x sparse product: too many workgroups for one launch");
	dim3 grid((unsigned) (npan * nkb));
	static int order = -1;
	if (order < 0) {
		order = 0;
#ifdef SVT_TUNING
		if (getenv("SVT_SPMM_ORDER")) order = atoi(getenv("SVT_SPMM_ORDER")) != 0;
#endif
	}
#define SPMM_GO(TA, TB) do { \
		(void) hipFuncSetAttribute((const void *) spmm_csc_csc_kernel<TA, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
		hipLaunchKernelGGL((spmm_csc_csc_kernel<TA, TB>), grid, dim3(g_spmm_nt), lds, s, a, KW, G, (int) nkb, order); } while (0)
	if (a.a_type == SVT_REALSXP && a.b_type == SVT_REALSXP) SPMM_GO(double, double);
	else if (a.a_type == SVT_INTSXP && a.b_type == SVT_INTSXP) SPMM_GO(int, int);
	else if (a.a_type == SVT_REALSXP && a.b_type == SVT_INTSXP) SPMM_GO(double, int);
	else if (a.a_type == SVT_INTSXP && a.b_type == SVT_REALSXP) SPMM_GO(int, double);
	else return svt_set_error("sparse x sparse product: unsupported operand types");
#undef SPMM_GO
	HIP_TRY(hipGetLastError());
	return 0;
}
