/*
 * svt_hip.h -- C ABI of libsvt_hip.so, the MI355X (gfx950) backend for
 * SparseArray's SVT compute hot path.
 *
 * Plain C: pointers and sizes only, no R types, no torch types.  There are
 * two levels:
 *
 *   1. Host level -- one function per `.Call` entry point of the reference
 *      (src/R_init_SparseArray.c:94,121-134).  Arguments are host pointers.
 *      The function marshals the SVT leaves into the CSC device layout,
 *      runs the HIP kernels and copies the result back into the caller's
 *      buffer.  This is what the R package's C glue binds.
 *
 *   2. Device level -- operands already resident in HBM (uploaded once with
 *      svt_upload(), or wrapped around existing device buffers), kernels
 *      launched asynchronously on a caller-supplied HIP stream, no
 *      allocation and no synchronisation inside.  This is what bench.py and
 *      the multi-GPU driver use.
 *
 * Return convention (all int-returning functions):
 *     0   success
 *   < 0   (-1) error; message in svt_last_error().  The R glue turns it into
 *         error(), like the reference's own error() calls
 *         (e.g. src/SparseMatrix_mult.c:943-966).
 *   > 0   (SVT_UNSUPPORTED = 1) not supported HERE: the device kernels do not take
 *         this operand or operation -- 2^31 nonzeros or more in a transposition /
 *         aperm / colMedians / the second operand of the row-panel product (an SVT's
 *         total count is unbounded, R/SVT_SparseArray-class.R:13-23), too many strata
 *         for the row-statistics counters, an opcode the R API never sends
 *         (RANGE, SUM_X_X2, VAR2, SD2 for col / row statistics).  The reason is in
 *         svt_last_error(); nothing the caller relies on has been written.  The R glue
 *         answers it with the reference's own CPU body for that call
 *         (integration/svt_hip_glue.c, HIP_STATUS) -- the same thing it does when the
 *         library or the GPU is absent.  Inside the library a route that is refused
 *         falls back to another one where there is one (the sparse-aware crossprod to
 *         the dense-buffer route, the row-panel product to the transposition route);
 *         only what no device route takes surfaces as > 0.
 * Warning conditions ("NAs introduced by coercion of infinite values to
 * integers", src/SparseArray_matrixStats.c:278-280; "NAs produced by integer
 * overflow", src/rowsum_methods.c:122-123) are reported through the
 * `warn` / `ovflow` out-parameters so that the caller raises them on the R
 * thread after the call.
 *
 * Element types use R's SEXPTYPE codes.  Dense matrices are column-major
 * (R layout).  Missing values: NA_integer_ = INT_MIN, NA_real_ = the NaN with
 * low word 1954.
 */
#ifndef SVT_HIP_H
#define SVT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVT_UNSUPPORTED 1

#define SVT_LGLSXP 10
#define SVT_INTSXP 13
#define SVT_REALSXP 14

/* Opcodes of src/Rvector_summarization.h:12-34 */
#define SVT_OP_ANYNA            1
#define SVT_OP_COUNTNAS         2
#define SVT_OP_ANY              3
#define SVT_OP_ALL              4
#define SVT_OP_MIN              5
#define SVT_OP_MAX              6
#define SVT_OP_RANGE            7
#define SVT_OP_SUM              8
#define SVT_OP_PROD             9
#define SVT_OP_MEAN            10
#define SVT_OP_CENTERED_X2_SUM 11
#define SVT_OP_SUM_X_X2        12
#define SVT_OP_VAR1            13
#define SVT_OP_VAR2            14
#define SVT_OP_SD1             15
#define SVT_OP_SD2             16

/*
 * Host view of an SVT (Sparse Vector Tree, src/leaf_utils.h:10-31): the
 * tree flattened to its prod(dim[1..ndim-1]) leaves in depth-first order
 * (outermost dim slowest).  nzcount[j] == 0 is a NULL leaf / NULL subtree;
 * nzvals[j] == NULL with nzcount[j] > 0 is a lacunar leaf (all ones).
 * Replaces the (x_dim, x_type, x_SVT) argument triple of the .Call entry
 * points (src/SparseMatrix_mult.h:6-43, src/SparseArray_matrixStats.h:6-28,
 * src/rowsum_methods.h:6-36).
 */
typedef struct svt_view {
	int32_t Rtype;
	int32_t ndim;
	const int32_t *dim;
	int32_t svt_is_null;          /* x@SVT is NULL */
	int64_t nleaves;
	const int32_t *nzcount;
	const int32_t *const *nzoffs;
	const void *const *nzvals;
	int32_t na_background;        /* NaArray: the implicit value is NA, not zero (R/NaArray-class.R) */
} svt_view;

/* ---------------------------------------------------------------------- */
/* Library state                                                          */
/* ---------------------------------------------------------------------- */

/* Selects the HIP device for the calling process (one process per GPU).
   Returns 0, or -1 when no gfx950 device is usable.  Called implicitly with
   device 0 by the first host-level call. */
int svt_init(int device);
const char *svt_last_error(void);
/* "gfx950" etc. of the selected device, or "" before svt_init(). */
const char *svt_device_arch(void);

/*
 * Resident operands (off by default).  R code calls the entry points below over and over
 * on the same object, and each call marshals and uploads the whole tree again.  With a
 * byte limit > 0 the host-level entry points keep the device copy of every SVT operand
 * (plus what they derive from it: the panel-blocked layout, t(x)) up to that many bytes,
 * least recently used first out, and recognise the operand of a later call by a
 * fingerprint of its view: dims, type, and per leaf the host pointers, the count and
 * eight sampled (offset, value) pairs.  R vectors are not modified once shared; callers
 * that overwrite leaves in place must call svt_resident_clear().  This is the device-side
 * counterpart of the reference operating in place on host memory
 * (src/SVT_SparseArray_class.c:598-633 walks the leaves on every call, at no cost).
 */
int svt_resident_set_limit(size_t bytes);          /* 0 = off, frees everything */
void svt_resident_clear(void);
void svt_resident_stats(size_t *bytes, int64_t *entries, int64_t *hits, int64_t *misses);

/* ---------------------------------------------------------------------- */
/* 1. Host level: the .Call entry points                                   */
/* ---------------------------------------------------------------------- */

/* C_crossprod2_SVT_mat, src/SparseMatrix_mult.c:931-982.
   out: ncol(x) x (tr_y ? y_nrow : y_ncol) doubles.
   Types: the reference's entry point requires type(x) == typeof(y) and its R method coerces the
   integer operand of a mixed pair on the host first; svt_crossprod2_SVT_mat, _mat_SVT and
   svt_matmul_SVT_mat also take an integer operand next to a double one and widen it on the device
   (NA_integer_ -> NA_real_, as as.double() does). */
int svt_crossprod2_SVT_mat(const svt_view *x, const void *y, int y_nrow,
			   int y_ncol, int y_Rtype, int tr_y, double *out);
/* C_crossprod2_mat_SVT, src/SparseMatrix_mult.c:985-1034.
   out: (tr_x ? x_nrow : x_ncol) x ncol(y) doubles. */
int svt_crossprod2_mat_SVT(const void *x, int x_nrow, int x_ncol, int x_Rtype,
			   const svt_view *y, int tr_x, double *out);
/* C_crossprod2_SVT_SVT, src/SparseMatrix_mult.c:1037-1101. */
int svt_crossprod2_SVT_SVT(const svt_view *x, const svt_view *y, double *out);
/* C_crossprod1_SVT, src/SparseMatrix_mult.c:1104-1140. */
int svt_crossprod1_SVT(const svt_view *x, double *out);
/* x %*% y in one call (R/SparseMatrix-mult.R:195-215: the R methods transpose x on the host
   with C_transpose_2D_SVT, src/SparseArray_aperm.c:348-423, then call C_crossprod2_SVT_mat /
   C_crossprod2_SVT_SVT).  Here x is uploaded once and transposed on the device.
   out: nrow(x) x ncol(y) doubles, column-major.  Same checks and messages as the
   crossprod2 entry points. */
int svt_matmul_SVT_mat(const svt_view *x, const void *y, int y_nrow, int y_ncol,
		       int y_Rtype, double *out);
int svt_matmul_SVT_SVT(const svt_view *x, const svt_view *y, double *out);

/* tcrossprod(x) and tcrossprod(x, y) of SVT_SparseMatrix objects in one call (round 6).  The R methods
   (R/SparseMatrix-mult.R:165-193) are crossprod(t(x)) / crossprod(t(x), t(y)) with t() on the host
   (C_transpose_2D_SVT, then a second marshalling of the transposed trees); here x (and y) are uploaded as they are and
   transposed on the device -- and the sparse-aware kernel of svt_dev_crossprod_csc_csc wants the ROWS of t(x), i.e. x
   itself, so its side needs no transposition at all.  Same checks, messages, route choice and NA / NaN rules as
   svt_crossprod1_SVT / svt_crossprod2_SVT_SVT on the transposed operands.
   out: nrow(x) x nrow(x) resp. nrow(x) x nrow(y) doubles, column-major. */
int svt_tcrossprod1_SVT(const svt_view *x, double *out);
int svt_tcrossprod2_SVT_SVT(const svt_view *x, const svt_view *y, double *out);

/* colMedians(x, na.rm) of a 2-D SVT: .colMedians_SVT_SparseMatrix / .padded_median,
   R/SparseArray-matrixStats.R:690-784 -- pure R in the reference (one sort per leaf; its TODO
   at :690-691 asks for a .Call version).  Median of each column's nrow values, the implicit
   zeros included; any NA/NaN among the nonzeros gives NA_real_ unless na_rm, an empty column
   of a 0-row matrix NA_real_.  out: ncol(x) doubles.  rowMedians(x) is colMedians(t(x)) as
   in the reference (:802-815). */
int svt_colMedians_SVT(const svt_view *x, int na_rm, double *out);
/* rowMedians(x): the same on t(x), transposed on the device.  out: nrow(x) doubles. */
int svt_rowMedians_SVT(const svt_view *x, int na_rm, double *out);

/* C_summarize_SVT, src/SparseArray_summarization.c:112-142.  The result is
   left in out_d[0..1] or out_i[0..1] according to *out_Rtype. */
int svt_summarize_SVT(const svt_view *x, int opcode, int na_rm, double center,
		      double *out_d, int *out_i, int *out_Rtype, int *warn);

/* Result element type of a col/row stat (src/Rvector_summarization.c:97-165):
   SVT_LGLSXP/SVT_INTSXP -> int32 buffer, SVT_REALSXP -> double buffer. */
int svt_colStats_out_Rtype(int opcode, int in_Rtype);
/* C_colStats_SVT, src/SparseArray_matrixStats.c:234-284.
   out: prod(dim[dims..]) elements. */
int svt_colStats_SVT(const svt_view *x, int opcode, int na_rm, double center,
		     int dims, void *out, int *warn);
/* C_rowStats_SVT, src/SparseArray_matrixStats.c:1121-1205.
   center: NULL or prod(dim[0..dims-1]) doubles.  out: same length. */
int svt_rowStats_SVT(const svt_view *x, int opcode, int na_rm,
		     const double *center, int dims, void *out, int *warn);

/* C_rowsum_SVT / C_colsum_SVT, src/rowsum_methods.c:281-325, 363-401.
   group: 1-based, NA allowed.  out: ngroup x ncol (rowsum) or nrow x ngroup
   (colsum); int32 for SVT_INTSXP input, else double. */
int svt_rowsum_SVT(const svt_view *x, const int *group, int ngroup, int na_rm,
		   void *out, int *ovflow);
int svt_colsum_SVT(const svt_view *x, const int *group, int ngroup, int na_rm,
		   void *out, int *ovflow);
/* C_rowsum_dgCMatrix / C_colsum_dgCMatrix, src/rowsum_methods.c:328-356,
   404-439: the (x, i, p) slots of a dgCMatrix. */
int svt_rowsum_dgCMatrix(int nrow, int ncol, const double *xx, const int *xi,
			 const int *xp, const int *group, int ngroup,
			 int na_rm, double *out);
int svt_colsum_dgCMatrix(int nrow, int ncol, const double *xx, const int *xi,
			 const int *xp, const int *group, int ngroup,
			 int na_rm, double *out);

/* C_colMins_dgCMatrix / C_colMaxs_dgCMatrix / C_colRanges_dgCMatrix / C_colVars_dgCMatrix,
   src/sparseMatrix_utils.c:128-166, 205-223 (registered at src/R_init_SparseArray.c:49-52;
   R wrappers R/sparseMatrix-utils.R:300-330): column statistics of a dgCMatrix from its
   Dim, x and p slots (the i slot is not read).  out: ncol doubles; colRanges: ncol x 2
   column-major (mins, then maxs).  NA / NaN rules of min_double() etc. (:15-103): an NA gives
   NA_real_ unless na_rm, else a NaN gives NaN; a column with fewer than nrow stored entries
   starts from 0.  colVars is col_var() (:173-203): IEEE arithmetic throughout. */
int svt_colMins_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
			  int na_rm, double *out);
int svt_colMaxs_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
			  int na_rm, double *out);
int svt_colRanges_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
			    int na_rm, double *out);
int svt_colVars_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
			  int na_rm, double *out);

/* ---------------------------------------------------------------------- */
/* 2. Device level                                                         */
/* ---------------------------------------------------------------------- */

/*
 * CSC-like device layout of an SVT (the device-side analogue of
 * dump_SVT_to_CsparseMatrix_slots(), src/SVT_SparseArray_class.c:598-633):
 *   col_ptr int64[ncol+1], row_idx int32[nnz], val f64|i32[nnz]
 * with ncol = number of leaves, nrow = dim[0]; lacunar leaves are expanded
 * to explicit ones.  All pointers are device pointers.
 */
typedef struct svt_dev_csc {
	int32_t Rtype;
	int32_t owned;        /* buffers were allocated by svt_upload() */
	int64_t nrow;
	int64_t ncol;
	int64_t nnz;
	int64_t *col_ptr;
	int32_t *row_idx;
	void *val;
	int32_t na_background; /* NaArray operand: implicit entries are NA (col stats / summarization
	                          only; set by svt_upload(), 0 for wrapped buffers) */
} svt_dev_csc;

/* Marshal + H2D.  Returns NULL on error. */
svt_dev_csc *svt_upload(const svt_view *x);
/* Wrap device buffers owned by the caller (e.g. a torch allocation). */
svt_dev_csc *svt_wrap_device_csc(int Rtype, int64_t nrow, int64_t ncol,
				 int64_t nnz, int64_t *col_ptr,
				 int32_t *row_idx, void *val);
void svt_release(svt_dev_csc *h);

/*
 * crossprod(A, Y): out[c, k] = sum_r A[r, c] * Y[r, k], the kernel family
 * K1-K8 of the reference (src/SparseMatrix_mult.c:131-239) for all K dense
 * columns in one pass over A.
 *   Y      in_nrow x K, column-major with leading dimension ldY (doubles for
 *          a REALSXP A, int32 for an INTSXP A); or, with tr_y != 0, K x
 *          in_nrow column-major (ldY >= K), i.e. rows of Y are contiguous.
 *   out    element (c, k) is written at out[c * out_stride_c + k *
 *          out_stride_k]: (1, ncol) gives the column-major ncol x K result
 *          of C_crossprod2_SVT_mat, (K, 1) gives the K x ncol result of
 *          C_crossprod2_mat_SVT.
 *   ws     workspace of at least svt_dev_crossprod_ws_bytes() bytes.
 * Asynchronous on `stream` (a hipStream_t).
 */
size_t svt_dev_crossprod_ws_bytes(int64_t nrow, int64_t ncol, int K);
/* The two phases of svt_dev_crossprod_csc_dense(), callable separately so a
   dense operand can be prepared once and multiplied several times (and so
   that each phase can be timed): (1) stage Y into `ws` and evaluate the
   reference's per-column prescan predicates (src/SparseMatrix_mult.c:23-36);
   (2) the sparse x dense product proper, reading `ws`. */
int svt_dev_dense_prepare(const void *Y, int64_t ldY, int64_t nrow, int K,
			  int tr_y, int Rtype, void *ws, size_t ws_bytes,
			  void *stream);
int svt_dev_crossprod_prepared(const svt_dev_csc *A, const void *ws, int K,
			       double *out, int64_t out_stride_c,
			       int64_t out_stride_k, void *stream);
int svt_dev_crossprod_csc_dense(const svt_dev_csc *A, const void *Y,
				int64_t ldY, int K, int tr_y, double *out,
				int64_t out_stride_c, int64_t out_stride_k,
				void *ws, size_t ws_bytes, void *stream);

/*
 * Fast path of crossprod(A, Y) for f64 operands: a panel-blocked copy of A
 * ("PBC", see sparsearray_amd/csrc/kernels_mult_pbc.hip) built once per
 * sparse operand -- the device analogue of the reference's per-call leaf
 * "preprocessing" (src/SparseMatrix_mult.c:632-724) -- and a kernel that keeps
 * row panels of Y in LDS and per-column partial sums in registers.
 *   CBW   columns per wavefront (<= 64), WPB wavefronts per workgroup, logR = log2(rows per
 *         panel).  (0, 0, 0) = chosen by the operand's density: (40, 16, 7), the LDS-DMA kernel,
 *         or, below ~0.25 % density, (40, 4, 9), the gather kernels (rows of Y straight from L2).
 * svt_dev_pbc_build() allocates and synchronises (not for the launch path).
 * svt_dev_crossprod_pbc() has the semantics and the out-indexing of
 * svt_dev_crossprod_csc_dense() (A is needed for the general path that
 * takes over when Y holds NaN/Inf/NA); Y is f64.
 */
typedef struct svt_dev_pbc svt_dev_pbc;
svt_dev_pbc *svt_dev_pbc_build(const svt_dev_csc *A, int CBW, int WPB, int logR);
void svt_dev_pbc_release(svt_dev_pbc *P);
/* Layout buffers come from a stream-ordered pool of the library's own (one per device) that keeps up to 3 GiB of
   released memory for the next build; svt_dev_pbc_release() frees behind the work of every stream that has run a
   product with the layout (an event recorded on each of them at release time: no device-wide synchronisation, and
   nothing put on the stream per product; a stream destroyed before the handle is released makes the release
   synchronise the device instead -- a stream that has run a product with a layout should outlive the layout's
   handle, or the handle be released first: the release records on every stream it noted, and a stale
   hipStream_t is only as safe as the runtime's validation of it).  svt_dev_pbc_trim() hands everything the pools hold
   but no layout uses back to the driver -- e.g. before another allocator of the process needs the memory. */
void svt_dev_pbc_trim(void);
/* Device bytes held by a layout (records + tile table + flags). */
size_t svt_dev_pbc_bytes(const svt_dev_pbc *P);
size_t svt_dev_crossprod_pbc_ws_bytes(const svt_dev_pbc *P, int K);
int svt_dev_crossprod_pbc(const svt_dev_pbc *P, const svt_dev_csc *A,
			  const double *Y, int64_t ldY, int K, int tr_y,
			  double *out, int64_t out_stride_c,
			  int64_t out_stride_k, void *ws, size_t ws_bytes,
			  void *stream);

/* CUs the LDS-DMA product kernel leaves idle (default 0).  Its workgroups take a whole CU each, so a
   collective's kernels cannot start beside it; with n > 0 (a multiple of 8: n / 8 per XCD) the row splits are
   chosen so that at most 256 - n workgroups run -- for the multi-GPU driver, whose all-reduce of step i
   runs beside the product of step i + 1 (the reference has no counterpart: its OpenMP loop,
   src/SparseMatrix_mult.c:253-258, owns the cores it is given).  Process-wide; takes effect at the next
   product (a workspace sized by svt_dev_crossprod_pbc_ws_bytes() fits either setting). */
void svt_dev_pbc_set_spare_cus(int n);
int svt_dev_pbc_spare_cus(void);

/* Pacing of the gather product (very sparse operands, K a multiple of 128, >= 64 row panels): its grid is
   persistent, every XCD owns a range of rows, and a wavefront runs at most `dsync` row panels ahead of the
   slowest wavefront of its XCD that has started, so that the rows of the dense operand the XCD gathers stay
   in its L2; `spin` = polls after which a wavefront that waits in vain stops pacing itself (results never
   depend on the pacing).  Defaults (1, 256).  dsync < 0:
   the unpaced kernels (one launch per chunk of rows) run instead.  Process-wide; no reference counterpart
   (src/SparseMatrix_mult.c:131-152 walks leaf by leaf on the host). */
void svt_dev_pbc_set_gather_pacing(int dsync, int spin);

/* Products with many column blocks and no row split (A %*% Y on the layout of t(A)) are launched one round of
   workgroups at a time, so that every round starts aligned and the dense tile its workgroups stage streams
   through the XCDs' L2 once per round, and the last, partly filled round (54 of 256 CUs at BASELINE config 2b) is cut by
   rows into 256 / its size splits whose partial sums are added in split order (round 5); on = 2: per round, the last round
   whole; on = 0: one launch.  Default 1.  Process-wide, tuning / measurement. */
void svt_dev_pbc_set_round_launches(int on);

/* The same, restricted to the leaves from `first_col` on (rounded down to the kernel's block of
   16 * CBW columns): cells of earlier leaves are not written.  What the unary crossprod(x) needs:
   of dense column k only the leaves c >= k (compute_sym_dotprods_*, src/SparseMatrix_mult.c:
   263-296, computes ncol^2 / 2 dot products and mirrors them). */
int svt_dev_crossprod_pbc_from(const svt_dev_pbc *P, const svt_dev_csc *A,
			       const double *Y, int64_t ldY, int K, int tr_y,
			       double *out, int64_t out_stride_c,
			       int64_t out_stride_k, void *ws, size_t ws_bytes,
			       void *stream, int64_t first_col);

/* The two phases of svt_dev_crossprod_pbc() separately (so that each can be
   timed): phase 1 = the LDS-panel product kernel (partial sums into ws),
   phase 2 = deterministic sum of the partials into `out` + the general path
   when Y is not finite. */
int svt_dev_crossprod_pbc_phase(const svt_dev_pbc *P, const svt_dev_csc *A,
				const double *Y, int64_t ldY, int K, int tr_y,
				double *out, int64_t out_stride_c,
				int64_t out_stride_k, void *ws, size_t ws_bytes,
				void *stream, int phase);

/* col stats over segments of `inner` consecutive leaves each
   (dims > 1 => inner = prod(dim[1..dims-1])); out has ncol/inner elements of
   svt_colStats_out_Rtype().  warn_flag: device int, set to 1 on the
   "NAs introduced" condition (may be NULL). */
int svt_dev_colstats(const svt_dev_csc *A, int opcode, int na_rm,
		     double center, int64_t inner, void *out, int *warn_flag,
		     void *stream);

/* colMedians on the device (see svt_colMedians_SVT): out = ncol doubles (device);
   ws: svt_dev_colmedians_ws_bytes() bytes (two f64 key arrays + the sort's scratch). */
size_t svt_dev_colmedians_ws_bytes(int64_t nnz, int64_t ncol);
int svt_dev_colmedians(const svt_dev_csc *A, int na_rm, double *out, void *ws, size_t ws_bytes,
		       void *stream);

/* row sums: out[(j % inner) * nrow + r] = sum over the leaves j that map to
   that cell.  Every output cell is owned by one workgroup (LDS row panels, no
   memory atomics); ws: svt_dev_rowstats_ws_bytes() bytes. */
size_t svt_dev_rowstats_ws_bytes(int64_t nrow, int64_t ncol);
int svt_dev_rowsums(const svt_dev_csc *A, int na_rm, int64_t inner,
		    double *out, void *ws, size_t ws_bytes, void *stream);
/* The table of run bounds per row panel that svt_dev_rowsums() derives from the operand's offsets (a quarter
   of its time at BASELINE config 2) depends on the operand and `inner` only: svt_dev_rowsums_prepare() leaves
   it in `ws` once, svt_dev_rowsums_prepared() -- same operand, same `inner`, same `ws` -- uses it as it is. */
int svt_dev_rowsums_prepare(const svt_dev_csc *A, int64_t inner, void *ws, size_t ws_bytes, void *stream);
int svt_dev_rowsums_prepared(const svt_dev_csc *A, int na_rm, int64_t inner,
			     double *out, void *ws, size_t ws_bytes, void *stream);

/* rowsum(): out (ngroup x ncol, zeroed by the callee); group is a device
   array of nrow 1-based group ids (NA -> last group).  f64 input only at
   this level. */
int svt_dev_rowsum(const svt_dev_csc *A, const int *group, int ngroup,
		   int na_rm, double *out, void *stream);
/* The same for an (operand, grouping) pair used more than once: svt_dev_rowsum_prepare() writes the group of
   every nonzero -- a 16-bit 0-based id, NA -> the last group (src/rowsum_methods.c:51-54) -- into `gid`
   (svt_dev_rowsum_gid_bytes(A) bytes, device memory); svt_dev_rowsum_prepared() then streams 10 bytes per
   nonzero (value + id) and looks nothing up.  1 <= ngroup <= 20480 (a column's sums live in LDS). */
size_t svt_dev_rowsum_gid_bytes(const svt_dev_csc *A);
int svt_dev_rowsum_prepare(const svt_dev_csc *A, const int *group, int ngroup, void *gid, size_t gid_bytes,
			   void *stream);
int svt_dev_rowsum_prepared(const svt_dev_csc *A, const void *gid, int ngroup, int na_rm, double *out,
			    void *stream);

/* Thread control (C_get_num_procs / C_get_max_threads / C_set_max_threads,
   src/thread_control.c:47-66; R/thread-control.R sets the team size around every
   .Call and restores it).  The device kernels have no thread team to size: the
   library reports the host's processor count, remembers the value it is given
   and returns the previous one, so SparseArray.Call() keeps working unchanged.
   No HIP call is made. */
int svt_get_num_procs(void);
int svt_get_max_threads(void);
int svt_set_max_threads(int nthread);

/* t(A) for a 2-d operand, CSC -> CSC (device counterpart of transpose_2D_SVT,
   src/SparseArray_aperm.c:148-423; every `%*%` / tcrossprod starts with it,
   R/SparseMatrix-mult.R:165-206).  The caller provides the output arrays
   (out_col_ptr int64[nrow+1], out_row_idx int32[nnz], out_val like A's) and
   svt_dev_transpose_ws_bytes() bytes of workspace; entries of every output leaf
   come out in ascending offset order. */
size_t svt_dev_transpose_ws_bytes(int64_t nrow, int64_t nnz);
int svt_dev_transpose(const svt_dev_csc *A, int64_t *out_col_ptr, int32_t *out_row_idx,
		      void *out_val, void *ws, size_t ws_bytes, void *stream);

/* x %*% y for two sparse operands, y much sparser than a dense matrix (the `svt %*% svt2` of BASELINE config 3):
   out[r + k * ldo] = sum over the nonzeros (j, b) of column k of B of b * A[r, j] -- for finite operands the sum
   the reference forms per cell (C_crossprod2_SVT_SVT on t(x), src/SparseMatrix_mult.c:1037-1101: one operand's
   leaves expanded, the other's walked over them, :728-820), without the order of its additions (the same
   products, added as the lane groups get to them; exact for integer operands below 2^53).  A non-finite
   value or an NA in either operand changes what the reference computes (its dirty-leaf loops multiply the
   implicit zeros too): then `*not_finite` (device int, may be NULL; the second int of `ws` holds the same flag)
   is set and `out` must be recomputed by the dense route (svt_matmul_SVT_SVT does; a launch that finds the flag
   up already leaves `out` alone).  Otherwise every cell of the A->nrow x B->ncol result is written; ws:
   svt_dev_matmul_csc_csc_ws_bytes(A) bytes.  Asynchronous.
   What depends on A alone -- the table of run bounds and a look at all its values, one pass over the operand --
   can be done once per operand: svt_dev_matmul_csc_csc_prepare(A, ws) fills `ws`, which
   svt_dev_matmul_csc_csc_prepared() then only reads (plus its second int, the flag of the last product), as the
   panel-blocked layout serves crossprod(A, Y).  svt_dev_matmul_csc_csc() does the same work for ONE product and
   less of it: its table pass looks only at the leaves of A that no column of B refers to, the product kernel at
   every value it reads (all of B, the other leaves of A); the `ws` it leaves is not a prepared one. */
size_t svt_dev_matmul_csc_csc_ws_bytes(const svt_dev_csc *A);
int svt_dev_matmul_csc_csc(const svt_dev_csc *A, const svt_dev_csc *B, double *out, int64_t ldo,
			   void *ws, size_t ws_bytes, int *not_finite, void *stream);
int svt_dev_matmul_csc_csc_prepare(const svt_dev_csc *A, void *ws, size_t ws_bytes, void *stream);
int svt_dev_matmul_csc_csc_prepared(const svt_dev_csc *A, const svt_dev_csc *B, double *out, int64_t ldo,
				    void *ws, size_t ws_bytes, int *not_finite, void *stream);

/* crossprod(X, Y) for two sparse operands without a dense buffer (round 6; kernels_gram.hip):
   out[c + j * ldo] = sum over the rows r where X[r, c] and Y[r, j] are both nonzero of X[r, c] * Y[r, j] -- for
   finite operands the cell C_crossprod2_SVT_SVT / C_crossprod1_SVT form (src/SparseMatrix_mult.c:1037-1140: one
   operand's leaves expanded into a dense buffer, ALL leaves of the other walked over it, :728-887; K11-K13
   :263-296), without the multiply-adds against the buffer's zeros and without the order of its additions
   (exact for integer operands below 2^53).
     Xt    t(X) in the device layout (svt_dev_transpose): Xt->nrow = ncol(X), Xt->ncol = nrow(X) leaves.
     Y     nrow(X) x ncol(Y).
     sym   != 0: Y is X (Xt = t(Y)), the unary crossprod(x): the cells c <= j are formed and mirrored
           (compute_sym_dotprods_*, :827-873, writes out[k] and out[k * ncol] from one dot product).
   A non-finite value or an NA anywhere in either operand changes what the reference computes (its dirty-leaf
   loops multiply the implicit zeros too): `*not_finite` (device int, may be NULL; the first int of `ws` holds
   the same flag) is set and `out` must come from the dense-buffer route (svt_crossprod2_SVT_SVT /
   svt_crossprod1_SVT do that).  ws: svt_dev_crossprod_csc_csc_ws_bytes(Xt) bytes.  Asynchronous. */
size_t svt_dev_crossprod_csc_csc_ws_bytes(const svt_dev_csc *Xt);
int svt_dev_crossprod_csc_csc(const svt_dev_csc *Xt, const svt_dev_csc *Y, int sym, double *out, int64_t ldo,
			      void *ws, size_t ws_bytes, int *not_finite, void *stream);
/* Results up to `one_block_max` cells tall (<= 20400, the default: 160 KB of LDS; two workgroups per CU up to 10200
   cells; the symmetric form stops at 16384) keep a whole result column in one workgroup's LDS; taller ones are
   cut into panels of 2^log2_panel (<= 14; default 13) cells.  Negative / out-of-range arguments restore the
   defaults.  Process-wide; tests and tuning (workspaces sized before a change may be too small after it). */
void svt_dev_crossprod_csc_csc_set_panel(int one_block_max, int log2_panel);

/* The dense-buffer route of the same product on resident operands -- the reference's own form (one operand
   densified 128 or more columns at a time, the panel / general product kernels against each chunk; the Lpp / Rpp
   choice of src/SparseMatrix_mult.c:1077-1097; Y == X, the same handle: the unary form, ncol^2 / 2 dot products
   + mirror): what the host entry points fall back to when an operand is not finite, and the yardstick the
   sparse-aware kernel is measured against.  out: ncol(X) x ncol(Y) doubles, column-major (device).  Allocates
   its buffers and synchronises the device: not for a launch path. */
int svt_dev_crossprod_csc_csc_dense_buffer(const svt_dev_csc *X, const svt_dev_csc *Y, double *out);

/* Route choice of svt_crossprod2_SVT_SVT / svt_crossprod1_SVT: the sparse-aware kernel above is taken when its
   estimated time -- pairs of nonzeros that meet in a row (nnz(x) * nnz(y) / nrow, half of it for the unary form)
   at the measured gather rate, plus t(x) -- times `factor` is below that of the dense-buffer route (the reference's
   own Lpp_nops / Rpp_nops count of multiply-adds, src/SparseMatrix_mult.c:1077-1078, at the measured rate of the
   panel kernels).  factor < 0: never; 0: always; default 1.  Process-wide. */
void svt_sparse_crossprod_set_cost(double factor);

/* aperm(x, perm) for an N-d operand (C_aperm_SVT, src/SparseArray_aperm.c:935-970;
   R/SparseArray-aperm.R).  `dim` are the array's ndim extents (dim[0] = A->nrow,
   prod(dim[1..]) = A->ncol), `perm` is 1-based as in R.  Output: the CSC layout of
   the permuted array, prod(dim[perm[1..]]) + 1 column pointers and A->nnz entries
   (caller-allocated); 1 <= ndim <= 8.  Asynchronous on `stream`, except for permutations whose new
   leading axis is an old outer axis and whose second axis is the old rows (aperm(x, c(3, 1, 2))): the
   choice between the per-slab kernel and the key sort reads one counter back and synchronises the
   stream once. */
size_t svt_dev_aperm_ws_bytes(int64_t nnz, int ndim, const int64_t *dim);
int svt_dev_aperm(const svt_dev_csc *A, int ndim, const int64_t *dim, const int *perm,
		  int64_t *out_col_ptr, int32_t *out_row_idx, void *out_val,
		  void *ws, size_t ws_bytes, void *stream);
/* How many transpositions / permutations of this process took which route (diagnostics; the differential fuzzers
   print it): counts[0] t() bucketed, [1] t() by the key sort, [2] aperm leaf-preserving, [3] first two axes swapped,
   [4] slab form, [5] 3-d through an intermediate, [6] general (composed), [7] key sort with 32-bit keys, [8] key sort
   with 64-bit keys, [9] slab form refused at run time.  Steps of composed routes count too.  reset != 0 zeroes them. */
void svt_dev_aperm_route_counts(int64_t counts[10], int reset);

/* Host level: same, on host buffers (x->nleaves leaves in, out_* as above with
   nnz = sum of x->nzcount). */
int svt_aperm_SVT(const svt_view *x, const int *perm, int64_t *out_col_ptr,
		  int32_t *out_row_idx, void *out_val);
/* C_transpose_2D_SVT, src/SparseArray_aperm.c:395-423 (t() of an SVT_SparseMatrix,
   R/SparseArray-aperm.R:11-20; every tcrossprod() and the non-native row*() statistics start
   with it, R/SparseMatrix-mult.R:165-206, R/SparseArray-matrixStats.R:140-147): x is uploaded,
   transposed on the device (svt_dev_transpose) and t(x) comes back as its CSC layout --
   x->dim[0] + 1 column pointers and sum(x->nzcount) entries, ascending offsets inside every
   leaf.  The glue rebuilds the R leaves from it (integration/svt_hip_glue.c). */
int svt_transpose_2D_SVT(const svt_view *x, int64_t *out_col_ptr,
			 int32_t *out_row_idx, void *out_val);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_H */
