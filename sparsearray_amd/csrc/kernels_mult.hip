// crossprod(): sparse (CSC leaves) x dense, all K dense columns in one pass.
//
// Reference: the 13 OpenMP loops of src/SparseMatrix_mult.c:131-296 call one
// of the per-leaf dot products of src/SparseVec_dotprod.c once per (leaf,
// dense column) pair, with the dense columns in the OUTER loop
// (crossprod2_SVT_mat_double :385-431), so the sparse operand is streamed K
// times.  Here A is streamed once per 64-column tile of Y and every nonzero is
// applied to 64 dense columns at a time (lane = dense column).
//
// Pipeline of one call:
//   1. prep:   Y (column-major, R layout) -> Yt[r][k] row-major, K padded to
//              a multiple of 64, so that the 64 lanes of a wavefront read one
//              contiguous 512-byte run per nonzero.  The same pass evaluates
//              the reference's prescan predicates per dense column
//              (has_no_NaN_or_Inf / has_no_NA, :23-36): a count of non-finite
//              entries and an "R NA present" bit.
//   2. gather: one wavefront per (leaf, 64-column tile); (row, value) of the
//              leaf are wave-uniform scalar loads; acc[lane] += v * Yt[row][lane]
//              in ascending offset order -- the summation order of
//              _dotprod_doubleSV_finite_doubles (SparseVec_dotprod.c:28-43).
//
// NA / NaN / Inf semantics without a second code path.  The reference switches
// to _dotprod_doubleSV_doubles (:48-65) when the dense column is not finite;
// that routine also multiplies the implicit zeros of the leaf, so a non-finite
// y at a zero row turns the result into NaN, and any R NA (in the column or in
// the leaf) gives NA_real_.  Equivalently, per (leaf, column):
//     column has NA                      -> NA_real_
//     column not finite and leaf has NA  -> NA_real_
//     #non-finite y gathered < #non-finite in the column -> sum + NaN
// which only needs the per-column counters from step 1.  Integer operands
// (:73-114): NA in the column or in the leaf -> NA_real_.
//
// Roofline: HBM.  Algorithmic bytes = 12 per nonzero + 8*nrow*K (Y) +
// 8*ncol*K (out).
#include "svt_common.h"

#define KT 64   // dense columns per wavefront tile

struct ColFlags {
	int *nonfinite;   // [Kp] count of NaN/Inf/NA entries (double) or NA (int)
	int *has_na;      // [Kp]
};

static inline int64_t pad_k(int K) { return ((int64_t) K + KT - 1) / KT * KT; }

size_t crossprod_ws_bytes(int64_t nrow, int64_t ncol, int K)
{
	(void) ncol;
	const int64_t Kp = pad_k(K);
	return (size_t) (nrow > 0 ? nrow : 1) * Kp * 8 + (size_t) Kp * 8 + 256;
}

// ---- step 1 -------------------------------------------------------------------
// Tiled transpose through LDS: a 64x64 tile, read along rows of Y's storage
// (consecutive r), written along k.
template <typename T>
__global__ void __launch_bounds__(256)
prep_dense_kernel(const T *__restrict__ Y, int64_t ldY, int64_t nrow, int K,
		  int tr_y, double *__restrict__ Yt, int64_t Kp, ColFlags fl,
		  const int *__restrict__ run_flag)
{
	__shared__ double tile[64][65];
	if (run_flag != NULL && *run_flag == 0)
		return;   // fast path already produced the result
	const int k0 = blockIdx.y * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
	const bool is_dbl = sizeof(T) == 8;
	// (row tiles in a grid-stride loop: the grid stays small, so that the launch that finds
	// run_flag == 0 -- every product on the fast path -- costs ~1 us instead of ~8)
	for (int64_t r0 = (int64_t) blockIdx.x * 64; r0 < nrow; r0 += (int64_t) gridDim.x * 64) {
	if (r0 != (int64_t) blockIdx.x * 64) __syncthreads();        // tile[] is reused
	if (!tr_y) {
		// element (r, k) at Y[r + k*ldY]: lanes run along r
		for (int kk = ty; kk < 64; kk += 4) {
			const int64_t r = r0 + tx;
			const int k = k0 + kk;
			double d = 0.0;
			if (r < nrow && k < K) {
				const T v = Y[r + (int64_t) k * ldY];
				bool na, nf;
				if (is_dbl) { d = (double) v; nf = !svt_is_finite(d); na = svt_is_na(d); }
				else { na = nf = ((int) v == NA_INT); d = na ? svt_na_real() : (double) v; }
				if (nf) atomicAdd(fl.nonfinite + k, 1);
				if (na) fl.has_na[k] = 1;
			}
			tile[kk][tx] = d;
		}
		__syncthreads();
		for (int rr = ty; rr < 64; rr += 4) {
			const int64_t r = r0 + rr;
			if (r < nrow)
				Yt[r * Kp + k0 + tx] = tile[tx][rr];
		}
	} else {
		// element (r, k) at Y[k + r*ldY]: already row-contiguous
		for (int rr = ty; rr < 64; rr += 4) {
			const int64_t r = r0 + rr;
			const int k = k0 + tx;
			double d = 0.0;
			if (r < nrow && k < K) {
				const T v = Y[k + r * ldY];
				bool na, nf;
				if (is_dbl) { d = (double) v; nf = !svt_is_finite(d); na = svt_is_na(d); }
				else { na = nf = ((int) v == NA_INT); d = na ? svt_na_real() : (double) v; }
				if (nf) atomicAdd(fl.nonfinite + k, 1);
				if (na) fl.has_na[k] = 1;
			}
			if (r < nrow)
				Yt[r * Kp + k] = d;
		}
	}
	}
}

// The same step for column-major doubles in tiles of 64 rows x 128 dense columns: every thread keeps
// sixteen 16-byte loads in flight (two consecutive rows of one dense column each; a wavefront reads two
// 512-byte runs), and the tile leaves as whole 1 KiB rows of Yt, one row per wavefront store -- the
// gather product's operand, whose preparation is a tenth of its step at BASELINE config 4
// (1.28 GB in, 1.28 GB out per rank: 0.57 ms with the 64 x 64 tiles above).
// Preconditions (dense_prepare checks them): Y 16-byte aligned, ldY even, Kp a multiple of 128.
#define PREP_LD 129
__global__ void __launch_bounds__(256)
prep_dense_rows128_kernel(const double *__restrict__ Y, int64_t ldY, int64_t nrow, int K,
			  double *__restrict__ Yt, int64_t Kp, ColFlags fl, const int *__restrict__ run_flag)
{
	extern __shared__ double ptile[];             // [64 rows][PREP_LD]
	if (run_flag != NULL && *run_flag == 0)
		return;
	const int k0 = blockIdx.y * 128;
	const int lane2 = (threadIdx.x & 31) * 2, kq = threadIdx.x >> 5;          // 32 row pairs x 8 dense columns
	const int wrow = threadIdx.x >> 6, wk = (threadIdx.x & 63) * 2;
	for (int64_t r0 = (int64_t) blockIdx.x * 64; r0 < nrow; r0 += (int64_t) gridDim.x * 64) {
		if (r0 != (int64_t) blockIdx.x * 64) __syncthreads();            // ptile[] is reused
		const int64_t r = r0 + lane2;
		double2 v[16];
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const int k = k0 + kq + 8 * i;
			v[i] = make_double2(0.0, 0.0);
			if (k < K) {
				const double *src = Y + r + (int64_t) k * ldY;
				if (r + 1 < nrow) v[i] = *(const double2 *) src;
				else if (r < nrow) v[i].x = src[0];
			}
		}
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const int k = k0 + kq + 8 * i;
			const bool nf0 = !svt_is_finite(v[i].x), nf1 = !svt_is_finite(v[i].y);
			if (nf0 | nf1) {
				atomicAdd(fl.nonfinite + k, (int) nf0 + (int) nf1);
				if (svt_is_na(v[i].x) || svt_is_na(v[i].y)) fl.has_na[k] = 1;
			}
			ptile[lane2 * PREP_LD + kq + 8 * i] = v[i].x;
			ptile[(lane2 + 1) * PREP_LD + kq + 8 * i] = v[i].y;
		}
		__syncthreads();
#pragma unroll 4
		for (int rr = wrow; rr < 64; rr += 4) {
			if (r0 + rr < nrow) {
				const double *t = ptile + rr * PREP_LD + wk;
				*(double2 *) (Yt + (r0 + rr) * Kp + k0 + wk) = make_double2(t[0], t[1]);
			}
		}
	}
}

// ---- step 2 -------------------------------------------------------------------
// (the work of one workgroup: four leaves x 64 dense columns; bx = block of leaves)
template <typename T>
__device__ inline void crossprod_gather_block(const int64_t *__restrict__ col_ptr,
					      const int32_t *__restrict__ row_idx,
					      const T *__restrict__ val, int64_t ncol,
					      const double *__restrict__ Yt, int64_t Kp, int K,
					      ColFlags fl, double *__restrict__ out,
					      int64_t sc, int64_t sk, int64_t bx)
{
	const int lane = threadIdx.x & 63;
	const int64_t c = bx * 4 + (threadIdx.x >> 6);
	if (c >= ncol)
		return;
	const int k = blockIdx.y * KT + lane;
	const bool is_dbl = sizeof(T) == 8;
	const int64_t beg = col_ptr[c], end = col_ptr[c + 1];
	const double *__restrict__ ycol = Yt + blockIdx.y * KT + lane;
	double acc = 0.0;
	int nf_hit = 0;
	bool leaf_na = false;
	for (int64_t i = beg; i < end; i++) {
		const int32_t r = row_idx[i];   // wave-uniform -> scalar load
		const T v = val[i];
		const double y = ycol[(int64_t) r * Kp];
		double dv;
		if (is_dbl) {
			dv = (double) v;
			leaf_na |= svt_is_na(dv);
		} else {
			leaf_na |= ((int) v == NA_INT);
			dv = (double) v;
		}
		nf_hit += svt_is_finite(y) ? 0 : 1;
		acc = __dadd_rn(acc, __dmul_rn(dv, y));
	}
	if (k >= K)
		return;
	const int col_nf = fl.nonfinite[k];
	double res = acc;
	if (is_dbl) {
		if (col_nf > 0) {
			if (fl.has_na[k] || leaf_na) res = svt_na_real();
			else if (nf_hit < col_nf) res = acc + NAN;
		} else if (leaf_na) {
			res = svt_na_real();
		}
	} else {
		if (col_nf > 0 || leaf_na) res = svt_na_real();
	}
	out[c * sc + (int64_t) k * sk] = res;
}

// Blocks of four leaves in a grid-stride loop: behind every panel product the kernel is enqueued with a flag that
// is almost always clear, and a grid of one workgroup per block -- 250000 x 2 for the 1e6 leaves of t(A) at
// BASELINE config 2b -- took 0.106 ms to find that out (5 % of that product); the grid is capped at 2048 x K / 64.
template <typename T>
__global__ void __launch_bounds__(256)
crossprod_gather_kernel(const int64_t *__restrict__ col_ptr,
			const int32_t *__restrict__ row_idx,
			const T *__restrict__ val, int64_t ncol,
			const double *__restrict__ Yt, int64_t Kp, int K,
			ColFlags fl, double *__restrict__ out,
			int64_t sc, int64_t sk, const int *__restrict__ run_flag)
{
	if (run_flag != NULL && *run_flag == 0)
		return;
	const int64_t nb = (ncol + 3) / 4;
	for (int64_t bx = blockIdx.x; bx < nb; bx += gridDim.x)
		crossprod_gather_block<T>(col_ptr, row_idx, val, ncol, Yt, Kp, K, fl, out, sc, sk, bx);
}

static ColFlags flags_of(void *ws, int64_t nrow, int64_t Kp)
{
	ColFlags fl;
	fl.nonfinite = (int *) ((double *) ws + (nrow > 0 ? nrow : 1) * Kp);
	fl.has_na = fl.nonfinite + Kp;
	return fl;
}

static int dense_prepare(const CrossprodArgs &a, const int *run_flag, hipStream_t s, bool clear = true);
static int crossprod_prepared(const CrossprodArgs &a, const int *run_flag, hipStream_t s);

int launch_dense_prepare(const CrossprodArgs &a, hipStream_t s) { return dense_prepare(a, NULL, s); }

// *any = 1 if some dense column holds a non-finite entry (per-column counters of the staging pass)
__global__ void any_nonfinite_kernel(const int *__restrict__ nonfinite, int K, int *__restrict__ any)
{
	for (int k = threadIdx.x; k < K; k += blockDim.x)
		if (nonfinite[k] > 0) *any = 1;
}

// staging pass + the "dense operand is not finite" flag the panel kernels' phase 2 looks at
int launch_dense_prepare_flag(const CrossprodArgs &a, int *any, hipStream_t s)
{
	if (dense_prepare(a, NULL, s))
		return -1;
	if (a.K > 0) {
		ColFlags fl = flags_of(a.ws, a.nrow, pad_k(a.K));
		hipLaunchKernelGGL(any_nonfinite_kernel, dim3(1), dim3(64), 0, s, fl.nonfinite, a.K, any);
		HIP_TRY(hipGetLastError());
	}
	return 0;
}
int launch_crossprod_prepared(const CrossprodArgs &a, hipStream_t s) { return crossprod_prepared(a, NULL, s); }

// Both phases, skipped on the device when *flag == 0 (see kernels_mult_pbc.hip).  counters_cleared:
// whoever sets the flag has zeroed the 2 * Kp per-column counters (crossprod_general_counters) in an
// earlier launch, so that a product with a clean dense operand does not pay for a memset it never uses.
int launch_crossprod_general_if(const CrossprodArgs &a, const int *flag, bool counters_cleared, hipStream_t s)
{
	if (a.ncol <= 0 || a.K <= 0)
		return 0;
	if (dense_prepare(a, flag, s, !counters_cleared))
		return -1;
	return crossprod_prepared(a, flag, s);
}

int *crossprod_general_counters(void *ws, int64_t nrow, int K, int *n)
{
	const int64_t Kp = pad_k(K);
	*n = (int) (2 * Kp);
	return flags_of(ws, nrow, Kp).nonfinite;
}

static int dense_prepare(const CrossprodArgs &a, const int *run_flag, hipStream_t s, bool clear)
{
	if (a.K <= 0)
		return 0;
	const int64_t Kp = pad_k(a.K);
	if (a.ws_bytes < crossprod_ws_bytes(a.nrow, a.ncol, a.K))
		return svt_set_error("crossprod workspace too small");
	double *Yt = (double *) a.ws;
	ColFlags fl = flags_of(a.ws, a.nrow, Kp);
	if (clear)
		HIP_TRY(hipMemsetAsync(fl.nonfinite, 0, (size_t) Kp * 8, s));
	if (a.nrow > 0) {
		const int64_t ntile = (a.nrow + 63) / 64;
		dim3 grid((unsigned) (ntile < 2048 ? ntile : 2048), (unsigned) (Kp / 64));
		if (a.Rtype == SVT_REALSXP && !a.tr_y && Kp % 128 == 0 && a.ldY % 2 == 0 && a.nrow >= 4096 &&
		    ((uintptr_t) a.Y & 15) == 0) {
			static bool lds_ok = false;
			if (!lds_ok) {
				HIP_TRY(hipFuncSetAttribute((const void *) prep_dense_rows128_kernel,
							    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * PREP_LD * 8));
				lds_ok = true;
			}
			dim3 g2((unsigned) (ntile < 2048 ? ntile : 2048), (unsigned) (Kp / 128));
			hipLaunchKernelGGL(prep_dense_rows128_kernel, g2, dim3(256), 64 * PREP_LD * 8, s,
					   (const double *) a.Y, a.ldY, a.nrow, a.K, Yt, Kp, fl, run_flag);
		} else if (a.Rtype == SVT_REALSXP)
			hipLaunchKernelGGL(prep_dense_kernel<double>, grid, dim3(256), 0, s,
					   (const double *) a.Y, a.ldY, a.nrow, a.K, a.tr_y, Yt, Kp, fl, run_flag);
		else
			hipLaunchKernelGGL(prep_dense_kernel<int>, grid, dim3(256), 0, s,
					   (const int *) a.Y, a.ldY, a.nrow, a.K, a.tr_y, Yt, Kp, fl, run_flag);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

static int crossprod_prepared(const CrossprodArgs &a, const int *run_flag, hipStream_t s)
{
	if (a.ncol <= 0 || a.K <= 0)
		return 0;
	const int64_t Kp = pad_k(a.K);
	const double *Yt = (const double *) a.ws;
	ColFlags fl = flags_of(a.ws, a.nrow, Kp);
	// (gated launches find their flag clear almost always: a small grid; ungated ones fill the chip a few times over)
	const int64_t nblk = (a.ncol + 3) / 4, cap = run_flag != NULL ? 2048 : 65536;
	dim3 grid((unsigned) (nblk < cap ? nblk : cap), (unsigned) (Kp / KT));
	if (a.Rtype == SVT_REALSXP)
		hipLaunchKernelGGL(crossprod_gather_kernel<double>, grid, dim3(256), 0, s,
				   a.col_ptr, a.row_idx, (const double *) a.val, a.ncol,
				   Yt, Kp, a.K, fl, a.out, a.out_stride_c, a.out_stride_k, run_flag);
	else
		hipLaunchKernelGGL(crossprod_gather_kernel<int>, grid, dim3(256), 0, s,
				   a.col_ptr, a.row_idx, (const int *) a.val, a.ncol,
				   Yt, Kp, a.K, fl, a.out, a.out_stride_c, a.out_stride_k, run_flag);
	HIP_TRY(hipGetLastError());
	return 0;
}

int launch_crossprod_csc_dense(const CrossprodArgs &a, hipStream_t s)
{
	if (a.ncol <= 0 || a.K <= 0)
		return 0;
	if (launch_dense_prepare(a, s))
		return -1;
	return launch_crossprod_prepared(a, s);
}

// ---- "preprocessing": leaves -> dense columns (src/SparseVec.c:9-47) ------------
template <typename T>
__global__ void densify_kernel(const int64_t *__restrict__ col_ptr,
			       const int32_t *__restrict__ row_idx,
			       const T *__restrict__ val, int64_t nrow, int64_t c0,
			       int64_t nc, T *__restrict__ dense)
{
	const int lane = threadIdx.x & 63;
	const int64_t j = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (j >= nc)
		return;
	const int64_t beg = col_ptr[c0 + j], end = col_ptr[c0 + j + 1];
	for (int64_t k = beg + lane; k < end; k += SVT_WAVE)
		dense[j * nrow + row_idx[k]] = val[k];
}

int launch_densify(const int64_t *col_ptr, const int32_t *row_idx,
		   const void *val, int Rtype, int64_t nrow, int64_t c0,
		   int64_t nc, void *dense, hipStream_t s)
{
	if (nc <= 0 || nrow <= 0)
		return 0;
	const size_t esz = Rtype == SVT_REALSXP ? 8 : 4;
	HIP_TRY(hipMemsetAsync(dense, 0, (size_t) nrow * nc * esz, s));
	const unsigned nb = (unsigned) ((nc + 3) / 4);
	if (Rtype == SVT_REALSXP)
		hipLaunchKernelGGL(densify_kernel<double>, dim3(nb), dim3(256), 0, s, col_ptr,
				   row_idx, (const double *) val, nrow, c0, nc, (double *) dense);
	else
		hipLaunchKernelGGL(densify_kernel<int>, dim3(nb), dim3(256), 0, s, col_ptr,
				   row_idx, (const int *) val, nrow, c0, nc, (int *) dense);
	HIP_TRY(hipGetLastError());
	return 0;
}

// int32 -> f64 (NA_integer_ -> NA_real_): integer operands of large products take the f64
// panel kernels.  The reference multiplies and adds integer operands in double anyway
// (_dotprod_intSV_*, src/SparseVec_dotprod.c:73-114), and an NA becomes the one non-finite
// value that sends a product down the general path, where "NA anywhere in the dense column
// or the leaf -> NA_real_" is the rule for both types.
__global__ void int_to_f64_kernel(const int *__restrict__ in, int64_t n, double *__restrict__ out)
{
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int v = in[i];
	out[i] = v == NA_INT ? svt_na_real() : (double) v;
}

int launch_int_to_f64(const int *in, int64_t n, double *out, hipStream_t s)
{
	if (n <= 0) return 0;
	hipLaunchKernelGGL(int_to_f64_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, in, n, out);
	HIP_TRY(hipGetLastError());
	return 0;
}

// compute_sym_dotprods_* write out[k] and out[k*n] from one dot product
// (src/SparseMatrix_mult.c:263-296): keep the (i > j) value, mirror it.
__global__ void mirror_lower_kernel(double *out, int64_t n)
{
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	for (int64_t j = blockIdx.y; j < i; j += gridDim.y)
		out[j + i * n] = out[i + j * n];
}

int launch_mirror_lower(double *out, int64_t n, hipStream_t s)
{
	if (n <= 1)
		return 0;
	dim3 grid((unsigned) ((n + 255) / 256), (unsigned) (n < 1024 ? n : 1024));
	hipLaunchKernelGGL(mirror_lower_kernel, grid, dim3(256), 0, s, out, n);
	HIP_TRY(hipGetLastError());
	return 0;
}
