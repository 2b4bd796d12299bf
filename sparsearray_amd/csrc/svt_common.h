// Shared device/host helpers for the SVT HIP backend (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svt_hip.h"

#define SVT_WAVE 64
#define NA_INT (-2147483647 - 1)

// R's NA_real_: the NaN whose low word is 1954 (R arithmetic.c).  Device code
// builds it from bits so that the payload survives constant folding.
__host__ __device__ inline double svt_na_real()
{
	union { unsigned long long u; double d; } x;
	x.u = 0x7FF00000000007A2ULL;
	return x.d;
}
__host__ __device__ inline bool svt_is_na(double v)   // R_IsNA()
{
	union { double d; unsigned long long u; } x;
	x.d = v;
	return v != v && (unsigned int) (x.u & 0xFFFFFFFFu) == 1954u;
}
__host__ __device__ inline bool svt_is_finite(double v)   // R_FINITE()
{
	union { double d; unsigned long long u; } x;
	x.d = v;
	return ((x.u >> 52) & 0x7FF) != 0x7FF;
}

// ---- 64-lane wavefront reductions -----------------------------------------
__device__ inline double wave_sum(double v)
{
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_down(v, o, SVT_WAVE);
	return v;   // valid in lane 0
}
__device__ inline double wave_prod(double v)
{
	for (int o = 32; o > 0; o >>= 1)
		v *= __shfl_down(v, o, SVT_WAVE);
	return v;
}
__device__ inline long long wave_sum_ll(long long v)
{
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_down(v, o, SVT_WAVE);
	return v;
}
__device__ inline int wave_or(int v)
{
	for (int o = 32; o > 0; o >>= 1)
		v |= __shfl_down(v, o, SVT_WAVE);
	return v;
}
__device__ inline double wave_min(double v)
{
	for (int o = 32; o > 0; o >>= 1) {
		double t = __shfl_down(v, o, SVT_WAVE);
		v = t < v ? t : v;
	}
	return v;
}
__device__ inline double wave_max(double v)
{
	for (int o = 32; o > 0; o >>= 1) {
		double t = __shfl_down(v, o, SVT_WAVE);
		v = t > v ? t : v;
	}
	return v;
}
__device__ inline int wave_min_i(int v)
{
	for (int o = 32; o > 0; o >>= 1) {
		int t = __shfl_down(v, o, SVT_WAVE);
		v = t < v ? t : v;
	}
	return v;
}
__device__ inline int wave_max_i(int v)
{
	for (int o = 32; o > 0; o >>= 1) {
		int t = __shfl_down(v, o, SVT_WAVE);
		v = t > v ? t : v;
	}
	return v;
}

// Order-preserving map double <-> uint64 so that integer atomicMin/Max order
// doubles (no NaNs go through it).
__device__ inline unsigned long long f64_to_ordered(double v)
{
	unsigned long long u = (unsigned long long) __double_as_longlong(v);
	return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ inline double ordered_to_f64(unsigned long long u)
{
	u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFULL) : ~u;
	return __longlong_as_double((long long) u);
}

// ---- host-side plumbing (svt_hip.cpp) ----------------------------------------
int svt_set_error(const char *fmt, ...);
int svt_set_unsupported(const char *fmt, ...);     // status > 0 at the ABI: the caller runs its CPU body
void svt_clear_unsupported(void);
#define HIP_TRY(expr)                                                         \
	do {                                                                  \
		hipError_t e__ = (expr);                                      \
		if (e__ != hipSuccess)                                        \
			return svt_set_error("%s failed: %s (%s:%d)", #expr, \
					     hipGetErrorString(e__),         \
					     __FILE__, __LINE__);             \
	} while (0)

// kernel launchers implemented in the .hip files ---------------------------------
struct StatsArgs {
	const int64_t *col_ptr;
	const void *val;
	int Rtype;
	int64_t nseg;       // number of results
	int64_t inner;      // leaves per segment
	int64_t seg_len;    // inner * dim0: length of the virtual vector
	int opcode;
	int na_rm;
	double center;
	void *out;
	int *warn_flag;
	int na_bg;          // NaArray: implicit values are NAs (Rvector_summarization.c:1078-1106)
	int dgc;            // dgCMatrix flavour of var1 (src/sparseMatrix_utils.c:173-223): plain IEEE, no NA rule
};
int launch_colstats(const StatsArgs &a, int64_t nnz, hipStream_t s);
size_t colmedians_ws_bytes(int64_t nnz, int64_t ncol);
int launch_colmedians(const int64_t *col_ptr, const void *val, int Rtype, int64_t nrow, int64_t ncol,
		      int64_t nnz, int na_rm, double *out, void *ws, hipStream_t s);

struct RowStatsArgs {
	const int64_t *col_ptr;
	const int32_t *row_idx;
	const void *val;
	int Rtype;
	int64_t ncol;       // leaves
	int64_t nrow;       // dim0
	int64_t inner;      // prod(dim[1..dims-1])
	int64_t nstrata;
	int64_t out_len;    // inner * nrow
	int opcode;
	int na_rm;
	const double *center;   // device, or NULL
	void *out;              // device, out_len elements
	void *scratch;          // device scratch (see rowstats_scratch_bytes)
	int *warn_flag;
	int64_t nnz_hint;       // nonzeros of the operand (launch tuning only), 0 = unknown
	int na_bg;              // NaArray: implicit entries are NAs (SparseArray_matrixStats.c:756-1019)
	int table_mode;         // launch_rowstats_panel: 0 = build the table of run bounds and use it, 1 = build it only,
	                        // 2 = it is in `ws` already (same operand, same operation class)
};
size_t rowstats_scratch_bytes(int opcode, int out_Rtype, int64_t out_len);
int launch_rowstats(const RowStatsArgs &a, int64_t nnz, hipStream_t s);      // memory atomics
size_t rowstats_panel_ws_bytes(int64_t nrow, int64_t ncol);
int launch_rowstats_panel(const RowStatsArgs &a, void *ws, hipStream_t s);      // LDS row panels
void launch_rowpanel_table(const int64_t *col_ptr, const int32_t *row_idx, int64_t ncol, int64_t nnz_hint,
			   int64_t npan, int ps, int32_t *pt, hipStream_t s);
bool launch_rowpanel_table_scan(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
				int64_t ncol, int64_t nnz_hint, int64_t npan, int ps, int32_t *pt,
				const uint8_t *skip, int *flag, hipStream_t s);

// sparse x sparse product by row panels (kernels_spmm.hip)
struct SpmmArgs {
	const int64_t *a_ptr; const int32_t *a_idx; const void *a_val; int a_type; int64_t nrow, ninner;
	const int64_t *b_ptr; const int32_t *b_idx; const void *b_val; int b_type; int64_t K;
	double *out; int64_t ldo;           // out[r + k * ldo]
	const int32_t *pt; int64_t npan; int ps;
	int *flag;                          // set when a non-finite value / an NA took part: the result is not the reference's
};
size_t spmm_ws_bytes(int64_t nrow, int64_t ninner);
int launch_spmm_prepare(const SpmmArgs &a, int64_t a_nnz, void *ws, hipStream_t s);
int launch_spmm_prepare_for(const SpmmArgs &a, int64_t a_nnz, int64_t b_nnz, void *ws, hipStream_t s, int *zero2);
int launch_spmm_product(SpmmArgs a, int64_t a_nnz, int64_t b_nnz, const void *ws, hipStream_t s);

// sparse x sparse crossprod without a dense operand (kernels_gram.hip)
struct GramArgs {
	const int64_t *a_ptr; const int32_t *a_idx; const void *a_val; int a_type;   // t(X): nrow leaves of (column of X, value)
	int64_t nx, nrow;
	const int64_t *b_ptr; const int32_t *b_idx; const void *b_val; int b_type; int64_t ny;   // Y: ny leaves of (row, value)
	double *out; int64_t ldo;           // out[c + j * ldo]
	int sym;                            // Y is X: cells c <= j, then mirrored
	const int32_t *pt; int64_t npan; int ps;
	int *flag;
};
void gram_set_panel(int one_block_max, int log2_panel);
size_t gram_ws_bytes(int64_t nx, int64_t nrow, int64_t a_nnz);
int launch_gram(GramArgs a, int64_t a_nnz, int64_t b_nnz, void *ws, hipStream_t s);
int launch_gram_mirror(double *out, int64_t n, int64_t ld, hipStream_t s);
int launch_gram_pairs(const int64_t *a_ptr, int64_t nrow, double *out, hipStream_t s);

size_t transpose_ws_bytes(int64_t nrow, int64_t nnz);
int launch_transpose(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
		     int64_t nrow, int64_t ncol, int64_t nnz, int64_t *out_ptr, int32_t *out_idx,
		     void *out_val, void *ws, hipStream_t s);

void aperm_route_counts(int64_t *out, int reset);
size_t aperm_ws_bytes(int64_t nnz, const int64_t *dim, int ndim);
int launch_aperm(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
		 int64_t ncol, int64_t nnz, const int64_t *dim, int ndim, const int *perm,
		 int64_t *out_ptr, int32_t *out_idx, void *out_val, void *ws, hipStream_t s);

struct GroupSumArgs {
	const int64_t *col_ptr64;   // one of the two col_ptr flavours is set
	const int32_t *col_ptr32;
	const int32_t *row_idx;
	const void *val;
	int Rtype;
	int64_t nrow, ncol;
	const int *group;           // device
	int ngroup;
	int na_rm;
	void *out;                  // device, zeroed by the launcher
	void *scratch;              // int path: int64 sums + NA flags
	int *ovflow_flag;
};
size_t groupsum_scratch_bytes(int Rtype, int64_t out_len);
int launch_rowsum(const GroupSumArgs &a, hipStream_t s);
int launch_rowsum_lds(const GroupSumArgs &a, hipStream_t s);   // f64, ngroup <= 8192
int launch_rowsum_gid(const GroupSumArgs &a, int64_t nnz, uint16_t *gid, hipStream_t s);
int launch_rowsum_prepared(const GroupSumArgs &a, const uint16_t *gid, hipStream_t s);   // f64, ngroup * 8 <= LDS
int launch_colsum(const GroupSumArgs &a, hipStream_t s);

struct CrossprodArgs {
	const int64_t *col_ptr;
	const int32_t *row_idx;
	const void *val;
	int Rtype;               // REALSXP or INTSXP (A and Y agree)
	int64_t nrow, ncol;
	const void *Y;           // dense operand (device)
	int64_t ldY;
	int K;
	int tr_y;
	double *out;
	int64_t out_stride_c, out_stride_k;
	void *ws;
	size_t ws_bytes;
};
size_t crossprod_ws_bytes(int64_t nrow, int64_t ncol, int K);
int launch_dense_prepare(const CrossprodArgs &a, hipStream_t s);
int launch_crossprod_prepared(const CrossprodArgs &a, hipStream_t s);
int launch_crossprod_csc_dense(const CrossprodArgs &a, hipStream_t s);
// Scatter leaves [c0, c0+nc) of a CSC into a dense column-major nrow x nc
// matrix (the "preprocessing" of src/SparseMatrix_mult.c:632-724).
int launch_densify(const int64_t *col_ptr, const int32_t *row_idx,
		   const void *val, int Rtype, int64_t nrow, int64_t c0,
		   int64_t nc, void *dense, hipStream_t s);
// out[j, i] = out[i, j] for i > j (n x n, column-major)
int launch_mirror_lower(double *out, int64_t n, hipStream_t s);
void pbc_auto_layout(int64_t nrow, int64_t ncol, int64_t nnz, int *CBW, int *WPB, int *logR);
int launch_int_to_f64(const int *in, int64_t n, double *out, hipStream_t s);
