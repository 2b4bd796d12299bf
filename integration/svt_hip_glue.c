/*
 * svt_hip_glue.c -- the host shim an R maintainer adds to SparseArray's src/ so that the package's
 * .Call entry points run on an MI355X through libsvt_hip.so (include/svt_hip.h of this repository).
 *
 * Not built into an R package in this repository (no R in the image), but syntax- and type-checked against
 * the reference's own headers by tests/test_glue_compiles.py and compiled, linked and RUN on the CPU by
 * tests/test_glue_executes.py (functional test-only stand-in for R's C API under tests/r_api_standin/, the
 * svt_* symbols bound to the CPU oracle, all golden cases through the registered names).  It is the complete
 * text of the binding,
 * one function per entry point registered in src/R_init_SparseArray.c:41-43,94,121-134:
 *
 *   C_crossprod2_SVT_mat/7  C_crossprod2_mat_SVT/7  C_crossprod2_SVT_SVT/8  C_crossprod1_SVT/5
 *   C_colStats_SVT/9        C_rowStats_SVT/9        C_summarize_SVT/7
 *   C_rowsum_SVT/6          C_colsum_SVT/6          C_rowsum_dgCMatrix/4    C_colsum_dgCMatrix/4
 *   C_get_num_procs/0       C_get_max_threads/0     C_set_max_threads/1
 *   C_transpose_2D_SVT/3    C_aperm_SVT/4           (src/R_init_SparseArray.c:70-72)
 *   C_colMins_dgCMatrix/2   C_colMaxs_dgCMatrix/2   C_colRanges_dgCMatrix/2  C_colVars_dgCMatrix/2
 *                                                   (src/R_init_SparseArray.c:49-52)
 *
 * How it is wired in: the reference's bodies keep their code under a new name (suffix _cpu: a one-line
 * rename per function); this file provides the registered names.  Every function follows the same
 * steps: argument checks the reference makes before it allocates (they stay on the R side, with the
 * reference's own helpers) -> result allocation with the reference's helpers (dimnames included) ->
 * SVT flattened to an svt_view (pointers into R's vectors, nothing copied) -> one svt_* call ->
 * status < 0 becomes error(svt_last_error()), status > 0 ("not supported here": e.g. 2^31 nonzeros or more in a
 * transposition) hands the call to the reference's own body, the warn / ovflow flags become warning() AFTER the
 * compute, on the R thread (src/SparseArray_matrixStats.c:278-280, src/rowsum_methods.c:122-123).
 * libsvt_hip.so is dlopen()ed on first use; without it, or without an MI355X, every entry point
 * runs its _cpu body, so the package still loads anywhere.
 *
 * Helpers of the reference used below, by name (they are not restated here):
 *   _get_and_check_Rtype_from_Rstring, _get_and_check_na_background   src/argcheck_utils.c
 *   _new_Rmatrix0, _new_Rarray0                                        src/Rvector_utils.c
 *   _get_summarize_opcode, _make_SummarizeOp, _init_SummarizeResult,
 *   _make_SEXP_from_summarize_result                                   src/Rvector_summarization.c
 *   unzip_leaf                                                         src/leaf_utils.h:80-140
 *   alloc_ans / compute_colStats_ans_dim / compute_rowStats_ans_dim / check_dims / check_rowStats_center /
 *   propagate_colStats_dimnames / propagate_rowStats_dimnames (static in src/SparseArray_matrixStats.c:
 *   made extern, prototypes below)
 *   check_group (static in src/rowsum_methods.c: made extern)
 */
#include <Rdefines.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "svt_hip.h"
#include "argcheck_utils.h"
#include "Rvector_utils.h"
#include "Rvector_summarization.h"
#include "leaf_utils.h"
#include "SparseMatrix_mult.h"
#include "SparseArray_matrixStats.h"
#include "SparseArray_summarization.h"
#include "rowsum_methods.h"
#include "SparseArray_aperm.h"
#include "sparseMatrix_utils.h"
#include "thread_control.h"

/* Helpers that are `static` in the reference and that its maintainer makes extern for this file
   (src/SparseArray_matrixStats.c:33, 44, 54, 108, 127, 151, 1079; src/rowsum_methods.c:15). */
int check_dims(SEXP dims, int min, int max);
SEXP compute_colStats_ans_dim(SEXP x_dim, int dims);
SEXP compute_rowStats_ans_dim(SEXP x_dim, int ans_ndim);
SEXP alloc_ans(SEXPTYPE Rtype, SEXP ans_dim, R_xlen_t *out_incs);
void propagate_colStats_dimnames(SEXP ans, SEXP x_dimnames, int dims);
void propagate_rowStats_dimnames(SEXP ans, SEXP x_dimnames, int dims);
const double *check_rowStats_center(SEXP center, SEXP x_dim, int ans_ndim);
void check_group(SEXP group, int x_nrow, int ngroup);

/* the reference's bodies, renamed */
SEXP C_crossprod2_SVT_mat_cpu(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP C_crossprod2_mat_SVT_cpu(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP C_crossprod2_SVT_SVT_cpu(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP C_crossprod1_SVT_cpu(SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP C_colStats_SVT_cpu(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP C_rowStats_SVT_cpu(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP C_summarize_SVT_cpu(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP C_rowsum_SVT_cpu(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP C_colsum_SVT_cpu(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP C_rowsum_dgCMatrix_cpu(SEXP, SEXP, SEXP, SEXP);
SEXP C_colsum_dgCMatrix_cpu(SEXP, SEXP, SEXP, SEXP);
SEXP C_transpose_2D_SVT_cpu(SEXP, SEXP, SEXP);
SEXP C_aperm_SVT_cpu(SEXP, SEXP, SEXP, SEXP);
SEXP C_colMins_dgCMatrix_cpu(SEXP, SEXP);
SEXP C_colMaxs_dgCMatrix_cpu(SEXP, SEXP);
SEXP C_colRanges_dgCMatrix_cpu(SEXP, SEXP);
SEXP C_colVars_dgCMatrix_cpu(SEXP, SEXP);
SEXP C_get_num_procs_cpu(void);
SEXP C_get_max_threads_cpu(void);
SEXP C_set_max_threads_cpu(SEXP);

/* ------------------------------------------------------------------------------------------------ */
/* the library, loaded lazily                                                                       */
/* ------------------------------------------------------------------------------------------------ */
static void *hip_lib;           /* NULL: not tried yet; (void *) -1: unavailable */
#define HIP_FN(name) ((__typeof__(&name)) dlsym(hip_lib, #name))

static int hip_available(void)
{
	if (hip_lib == NULL) {
		/* options(SparseArray.device = NA) / SPARSEARRAY_HIP_LIB="" keep the CPU path */
		const char *path = getenv("SPARSEARRAY_HIP_LIB");
		if (path != NULL && path[0] == '\0') {
			hip_lib = (void *) -1;
		} else {
			hip_lib = dlopen(path ? path : "libsvt_hip.so", RTLD_NOW | RTLD_LOCAL);
			if (hip_lib == NULL || HIP_FN(svt_init)(0) != 0)
				hip_lib = (void *) -1;
		}
	}
	return hip_lib != (void *) -1;
}

static void hip_fail(void)     /* same role as the reference's error() calls: never returns */
{
	error("%s", HIP_FN(svt_last_error)());
}

/* Status of an svt_* call (include/svt_hip.h): 0 done; < 0 error -> error(), as the reference's own error() calls;
   > 0 "not supported here" (an operand past a size limit of the device kernels, an operation they do not
   implement): what was PROTECTed for the device call is released and the reference's body computes the call. */
#define HIP_STATUS(call, nprotect, cpu_call)        \
	do {                                        \
		int rc__ = (call);                  \
		if (rc__ < 0)                       \
			hip_fail();                 \
		if (rc__ > 0) {                     \
			UNPROTECT(nprotect);        \
			return cpu_call;            \
		}                                   \
	} while (0)

static int device_type(SEXPTYPE Rtype)
{
	return Rtype == REALSXP || Rtype == INTSXP || Rtype == LGLSXP;
}

/* ------------------------------------------------------------------------------------------------ */
/* (x_dim, x_type, x_SVT) -> svt_view                                                               */
/* ------------------------------------------------------------------------------------------------ */
/* Depth-first walk in the order of REC_colStats_SVT (src/SparseArray_matrixStats.c:200-231): one
   slot per leaf; a NULL subtree leaves its prod(dim[1..ndim-1]) slots at nzcount == 0. */
static void fill_leaf_table(SEXP SVT, const int *dim, int ndim, R_xlen_t *pos,
			    int *nzcount, const int **nzoffs, const void **nzvals)
{
	if (ndim == 1) {
		if (SVT != R_NilValue) {
			SEXP vals, offs;
			nzcount[*pos] = unzip_leaf(SVT, &vals, &offs);
			nzoffs[*pos] = INTEGER(offs);
			nzvals[*pos] = vals == R_NilValue ? NULL : DATAPTR(vals);   /* lacunar leaf */
		}
		(*pos)++;
		return;
	}
	if (SVT == R_NilValue) {
		R_xlen_t n = 1;
		for (int a = 1; a < ndim; a++)
			n *= dim[a];
		*pos += n;
		return;
	}
	for (int i = 0; i < dim[ndim - 1]; i++)
		fill_leaf_table(VECTOR_ELT(SVT, i), dim, ndim - 1, pos, nzcount, nzoffs, nzvals);
}

/* The table lives in R_alloc() memory: released when the .Call returns, like the reference's scratch. */
static svt_view make_view(SEXP x_dim, SEXPTYPE Rtype, SEXP x_SVT, int na_background)
{
	svt_view v;
	int ndim = LENGTH(x_dim);
	R_xlen_t n = 1, pos = 0;
	for (int a = 1; a < ndim; a++)
		n *= INTEGER(x_dim)[a];
	size_t m = n > 0 ? (size_t) n : 1;
	int *cnt = (int *) R_alloc(m, sizeof(int));
	const int **offs = (const int **) R_alloc(m, sizeof(int *));
	const void **vals = (const void **) R_alloc(m, sizeof(void *));
	memset(cnt, 0, m * sizeof(int));
	memset(offs, 0, m * sizeof(int *));
	memset(vals, 0, m * sizeof(void *));
	fill_leaf_table(x_SVT, INTEGER(x_dim), ndim, &pos, cnt, offs, vals);
	v.Rtype = Rtype;
	v.ndim = ndim;
	v.dim = INTEGER(x_dim);
	v.svt_is_null = x_SVT == R_NilValue;
	v.nleaves = n;
	v.nzcount = cnt;
	v.nzoffs = offs;
	v.nzvals = vals;
	v.na_background = na_background;
	return v;
}

static void check_real_ans_type(SEXP ans_type, const char *fun)
{
	SEXPTYPE t = _get_and_check_Rtype_from_Rstring(ans_type, fun, "ans_type");
	if (t != REALSXP)
		error("SparseArray internal error in %s():\n"
		      "    output type \"%s\" is not supported yet", fun, type2char(t));
}

/* ------------------------------------------------------------------------------------------------ */
/* crossprod family -- src/SparseMatrix_mult.c:931-1140                                             */
/* ------------------------------------------------------------------------------------------------ */
SEXP C_crossprod2_SVT_mat(SEXP x_dim, SEXP x_type, SEXP x_SVT, SEXP y, SEXP transpose_y,
			  SEXP ans_type, SEXP ans_dimnames)
{
	if (!hip_available())
		return C_crossprod2_SVT_mat_cpu(x_dim, x_type, x_SVT, y, transpose_y, ans_type, ans_dimnames);
	int tr_y = LOGICAL(transpose_y)[0];
	SEXP y_dim = GET_DIM(y);
	if (LENGTH(x_dim) != 2 || LENGTH(y_dim) != 2)
		error("input objects must have 2 dimensions");
	SEXPTYPE Rtype = _get_and_check_Rtype_from_Rstring(x_type, "C_crossprod2_SVT_mat", "x_type");
	check_real_ans_type(ans_type, "C_crossprod2_SVT_mat");
	int y_nrow = INTEGER(y_dim)[0], y_ncol = INTEGER(y_dim)[1];
	/* conformability and the type pair are checked by the library with the reference's messages;
	   the result must not be allocated before they pass */
	if (INTEGER(x_dim)[0] != (tr_y ? y_ncol : y_nrow))
		error("input objects are non-conformable");
	SEXP ans = PROTECT(_new_Rmatrix0(REALSXP, INTEGER(x_dim)[1], tr_y ? y_nrow : y_ncol, ans_dimnames));
	svt_view xv = make_view(x_dim, Rtype, x_SVT, 0);
	HIP_STATUS(HIP_FN(svt_crossprod2_SVT_mat)(&xv, DATAPTR(y), y_nrow, y_ncol, TYPEOF(y), tr_y, REAL(ans)), 1,
		   C_crossprod2_SVT_mat_cpu(x_dim, x_type, x_SVT, y, transpose_y, ans_type, ans_dimnames));
	UNPROTECT(1);
	return ans;
}

SEXP C_crossprod2_mat_SVT(SEXP x, SEXP y_dim, SEXP y_type, SEXP y_SVT, SEXP transpose_x,
			  SEXP ans_type, SEXP ans_dimnames)
{
	if (!hip_available())
		return C_crossprod2_mat_SVT_cpu(x, y_dim, y_type, y_SVT, transpose_x, ans_type, ans_dimnames);
	int tr_x = LOGICAL(transpose_x)[0];
	SEXP x_dim = GET_DIM(x);
	if (LENGTH(x_dim) != 2 || LENGTH(y_dim) != 2)
		error("input objects must have 2 dimensions");
	SEXPTYPE Rtype = _get_and_check_Rtype_from_Rstring(y_type, "C_crossprod2_mat_SVT", "y_type");
	check_real_ans_type(ans_type, "C_crossprod2_mat_SVT");
	int x_nrow = INTEGER(x_dim)[0], x_ncol = INTEGER(x_dim)[1];
	if ((tr_x ? x_ncol : x_nrow) != INTEGER(y_dim)[0])
		error("input objects are non-conformable");
	SEXP ans = PROTECT(_new_Rmatrix0(REALSXP, tr_x ? x_nrow : x_ncol, INTEGER(y_dim)[1], ans_dimnames));
	svt_view yv = make_view(y_dim, Rtype, y_SVT, 0);
	HIP_STATUS(HIP_FN(svt_crossprod2_mat_SVT)(DATAPTR(x), x_nrow, x_ncol, TYPEOF(x), &yv, tr_x, REAL(ans)), 1,
		   C_crossprod2_mat_SVT_cpu(x, y_dim, y_type, y_SVT, transpose_x, ans_type, ans_dimnames));
	UNPROTECT(1);
	return ans;
}

SEXP C_crossprod2_SVT_SVT(SEXP x_dim, SEXP x_type, SEXP x_SVT, SEXP y_dim, SEXP y_type, SEXP y_SVT,
			  SEXP ans_type, SEXP ans_dimnames)
{
	if (!hip_available())
		return C_crossprod2_SVT_SVT_cpu(x_dim, x_type, x_SVT, y_dim, y_type, y_SVT, ans_type, ans_dimnames);
	if (LENGTH(x_dim) != 2 || LENGTH(y_dim) != 2)
		error("input objects must have 2 dimensions");
	if (INTEGER(x_dim)[0] != INTEGER(y_dim)[0])
		error("input SVT_SparseMatrix objects are non-conformable");
	SEXPTYPE x_Rtype = _get_and_check_Rtype_from_Rstring(x_type, "C_crossprod2_SVT_SVT", "x_type");
	SEXPTYPE y_Rtype = _get_and_check_Rtype_from_Rstring(y_type, "C_crossprod2_SVT_SVT", "y_type");
	if (x_Rtype != y_Rtype)
		error("input SVT_SparseMatrix objects must have the same type() for now");
	check_real_ans_type(ans_type, "C_crossprod2_SVT_SVT");
	SEXP ans = PROTECT(_new_Rmatrix0(REALSXP, INTEGER(x_dim)[1], INTEGER(y_dim)[1], ans_dimnames));
	svt_view xv = make_view(x_dim, x_Rtype, x_SVT, 0);
	svt_view yv = make_view(y_dim, y_Rtype, y_SVT, 0);
	HIP_STATUS(HIP_FN(svt_crossprod2_SVT_SVT)(&xv, &yv, REAL(ans)), 1,
		   C_crossprod2_SVT_SVT_cpu(x_dim, x_type, x_SVT, y_dim, y_type, y_SVT, ans_type, ans_dimnames));
	UNPROTECT(1);
	return ans;
}

SEXP C_crossprod1_SVT(SEXP x_dim, SEXP x_type, SEXP x_SVT, SEXP ans_type, SEXP ans_dimnames)
{
	if (!hip_available())
		return C_crossprod1_SVT_cpu(x_dim, x_type, x_SVT, ans_type, ans_dimnames);
	if (LENGTH(x_dim) != 2)
		error("'x' must have 2 dimensions");
	SEXPTYPE Rtype = _get_and_check_Rtype_from_Rstring(x_type, "C_crossprod1_SVT", "x_type");
	check_real_ans_type(ans_type, "C_crossprod1_SVT");
	int n = INTEGER(x_dim)[1];
	SEXP ans = PROTECT(_new_Rmatrix0(REALSXP, n, n, ans_dimnames));
	svt_view xv = make_view(x_dim, Rtype, x_SVT, 0);
	HIP_STATUS(HIP_FN(svt_crossprod1_SVT)(&xv, REAL(ans)), 1,
		   C_crossprod1_SVT_cpu(x_dim, x_type, x_SVT, ans_type, ans_dimnames));
	UNPROTECT(1);
	return ans;
}

/* ------------------------------------------------------------------------------------------------ */
/* matrixStats -- src/SparseArray_matrixStats.c:234-284, 1121-1205                                  */
/* ------------------------------------------------------------------------------------------------ */
static SEXPTYPE sexptype_of(int svt_Rtype)
{
	return svt_Rtype == SVT_REALSXP ? REALSXP : svt_Rtype == SVT_INTSXP ? INTSXP : LGLSXP;
}

/* Result objects exactly as the reference's entry points make them (src/SparseArray_matrixStats.c:
   255-266 and 1143-1150): alloc_ans() + the dimnames helper, with the reference's own functions. */
static SEXP alloc_colStats_ans(SEXPTYPE Rtype, SEXP x_dim, SEXP x_dimnames, int d)
{
	SEXP ans_dim = PROTECT(compute_colStats_ans_dim(x_dim, d));
	int ans_ndim = LENGTH(ans_dim);
	R_xlen_t *incs = ans_ndim != 0 ? (R_xlen_t *) R_alloc(ans_ndim, sizeof(R_xlen_t)) : NULL;
	SEXP ans = PROTECT(alloc_ans(Rtype, ans_dim, incs));
	propagate_colStats_dimnames(ans, x_dimnames, d);
	UNPROTECT(2);
	return ans;
}

static SEXP alloc_rowStats_ans(SEXPTYPE Rtype, SEXP ans_dim, SEXP x_dimnames, int ans_ndim)
{
	R_xlen_t *incs = (R_xlen_t *) R_alloc(ans_ndim, sizeof(R_xlen_t));       /* ans_ndim >= 1 */
	SEXP ans = PROTECT(alloc_ans(Rtype, ans_dim, incs));
	propagate_rowStats_dimnames(ans, x_dimnames, ans_ndim);
	UNPROTECT(1);
	return ans;
}

SEXP C_colStats_SVT(SEXP x_dim, SEXP x_dimnames, SEXP x_type, SEXP x_SVT, SEXP x_na_background,
		    SEXP op, SEXP na_rm, SEXP center, SEXP dims)
{
	SEXPTYPE Rtype = _get_and_check_Rtype_from_Rstring(x_type, "C_colStats_SVT", "x_type");
	if (!hip_available() || !device_type(Rtype))     /* complex / character / raw / list stay on the CPU */
		return C_colStats_SVT_cpu(x_dim, x_dimnames, x_type, x_SVT, x_na_background, op, na_rm,
					  center, dims);
	int na_bg = _get_and_check_na_background(x_na_background, "C_colStats_SVT", "x_na_background");
	int opcode = _get_summarize_opcode(op, Rtype);
	if (!(IS_LOGICAL(na_rm) && LENGTH(na_rm) == 1))
		error("'na.rm' must be TRUE or FALSE");
	if (!IS_NUMERIC(center) || LENGTH(center) != 1)
		error("SparseArray internal error in C_colStats_SVT():\n"
		      "    'center' must be a single number");
	int d = check_dims(dims, 1, LENGTH(x_dim));
	int warn = 0;
	SEXPTYPE ans_Rtype = sexptype_of(HIP_FN(svt_colStats_out_Rtype)(opcode, Rtype));
	/* result = array over tail(dim, -dims) with the matching dimnames: the reference's own
	   allocation path (alloc_ans() + propagate_colStats_dimnames(), :108-176) */
	SEXP ans = PROTECT(alloc_colStats_ans(ans_Rtype, x_dim, x_dimnames, d));
	svt_view xv = make_view(x_dim, Rtype, x_SVT, na_bg);
	HIP_STATUS(HIP_FN(svt_colStats_SVT)(&xv, opcode, LOGICAL(na_rm)[0], REAL(center)[0], d, DATAPTR(ans), &warn), 1,
		   C_colStats_SVT_cpu(x_dim, x_dimnames, x_type, x_SVT, x_na_background, op, na_rm, center, dims));
	if (warn)
		warning("NAs introduced by coercion of infinite values to integers");
	UNPROTECT(1);
	return ans;
}

SEXP C_rowStats_SVT(SEXP x_dim, SEXP x_dimnames, SEXP x_type, SEXP x_SVT, SEXP x_na_background,
		    SEXP op, SEXP na_rm, SEXP center, SEXP dims)
{
	SEXPTYPE Rtype = _get_and_check_Rtype_from_Rstring(x_type, "C_rowStats_SVT", "x_type");
	if (!hip_available() || !device_type(Rtype))
		return C_rowStats_SVT_cpu(x_dim, x_dimnames, x_type, x_SVT, x_na_background, op, na_rm,
					  center, dims);
	int na_bg = _get_and_check_na_background(x_na_background, "C_rowStats_SVT", "x_na_background");
	int opcode = _get_summarize_opcode(op, Rtype);
	if (!(IS_LOGICAL(na_rm) && LENGTH(na_rm) == 1))
		error("'na.rm' must be TRUE or FALSE");
	int ans_ndim = check_dims(dims, 1, LENGTH(x_dim) - 1);
	const double *center_p = check_rowStats_center(center, x_dim, ans_ndim);   /* NULL or head(dim, dims) doubles */
	int warn = 0;
	SEXPTYPE ans_Rtype = sexptype_of(HIP_FN(svt_colStats_out_Rtype)(opcode, Rtype));
	SEXP ans_dim = PROTECT(compute_rowStats_ans_dim(x_dim, ans_ndim));
	SEXP ans = PROTECT(alloc_rowStats_ans(ans_Rtype, ans_dim, x_dimnames, ans_ndim));
	svt_view xv = make_view(x_dim, Rtype, x_SVT, na_bg);
	/* ops the device does not cover natively for this shape come back as status > 0: CPU body */
	HIP_STATUS(HIP_FN(svt_rowStats_SVT)(&xv, opcode, LOGICAL(na_rm)[0], center_p, ans_ndim, DATAPTR(ans), &warn), 2,
		   C_rowStats_SVT_cpu(x_dim, x_dimnames, x_type, x_SVT, x_na_background, op, na_rm, center, dims));
	if (warn)
		warning("NAs introduced by coercion of infinite values to integers");
	UNPROTECT(2);
	return ans;
}

/* src/SparseArray_summarization.c:112-142 */
SEXP C_summarize_SVT(SEXP x_dim, SEXP x_type, SEXP x_SVT, SEXP x_na_background,
		     SEXP op, SEXP na_rm, SEXP center)
{
	SEXPTYPE Rtype = _get_and_check_Rtype_from_Rstring(x_type, "C_summarize_SVT", "x_type");
	if (!hip_available() || !device_type(Rtype))
		return C_summarize_SVT_cpu(x_dim, x_type, x_SVT, x_na_background, op, na_rm, center);
	int na_bg = _get_and_check_na_background(x_na_background, "C_summarize_SVT", "x_na_background");
	int opcode = _get_summarize_opcode(op, Rtype);
	if (!(IS_LOGICAL(na_rm) && LENGTH(na_rm) == 1))
		error("'na.rm' must be TRUE or FALSE");
	if (!IS_NUMERIC(center) || LENGTH(center) != 1)
		error("SparseArray internal error in C_summarize_SVT():\n"
		      "    'center' must be a single number");
	svt_view xv = make_view(x_dim, Rtype, x_SVT, na_bg);
	double out_d[2] = {0.0, 0.0};
	int out_i[2] = {0, 0}, out_Rtype = 0, warn = 0;
	HIP_STATUS(HIP_FN(svt_summarize_SVT)(&xv, opcode, LOGICAL(na_rm)[0], REAL(center)[0],
					     out_d, out_i, &out_Rtype, &warn), 0,
		   C_summarize_SVT_cpu(x_dim, x_type, x_SVT, x_na_background, op, na_rm, center));
	if (warn)
		warning("NAs introduced by coercion of infinite values to integers");
	/* The library hands back the post-processed state (value(s) + their type); the R object is made from it
	   by the reference's own _make_SEXP_from_summarize_result() (src/Rvector_summarization.c:1243-1303:
	   sum / prod of integers -> integer when it fits, countNAs -> integer, any / all / anyNA -> logical,
	   range -> 2 values) so that the two paths cannot disagree on the result's type. */
	SummarizeOp sop = _make_SummarizeOp(opcode, Rtype, LOGICAL(na_rm)[0], REAL(center)[0]);
	SummarizeResult res;
	_init_SummarizeResult(&sop, &res);
	res.out_Rtype = sexptype_of(out_Rtype);
	res.outbuf_status = OUTBUF_IS_SET;
	if (out_Rtype == SVT_REALSXP) {
		res.outbuf.two_doubles[0] = out_d[0];
		res.outbuf.two_doubles[1] = out_d[1];
	} else {
		res.outbuf.two_ints[0] = out_i[0];
		res.outbuf.two_ints[1] = out_i[1];
	}
	return _make_SEXP_from_summarize_result(&sop, &res);
}

/* ------------------------------------------------------------------------------------------------ */
/* rowsum / colsum -- src/rowsum_methods.c:281-439                                                  */
/* ------------------------------------------------------------------------------------------------ */
static SEXP groupsum_SVT(SEXP x_dim, SEXP x_type, SEXP x_SVT, SEXP group, SEXP ngroup, SEXP na_rm,
			 int colsum, const char *fun)
{
	if (LENGTH(x_dim) != 2)
		error("input object must have 2 dimensions");
	int x_nrow = INTEGER(x_dim)[0], x_ncol = INTEGER(x_dim)[1];
	SEXPTYPE Rtype = _get_and_check_Rtype_from_Rstring(x_type, fun, "x_type");
	int ng = INTEGER(ngroup)[0];
	check_group(group, colsum ? x_ncol : x_nrow, ng);
	if ((double) ng * (double) (colsum ? x_nrow : x_ncol) > 2147483647.0)
		error("too many groups (matrix of sums will be too big)");
	if (Rtype != REALSXP && Rtype != INTSXP)
		error("rowsum() and colsum() do not support SVT_SparseMatrix objects of\n"
		      "  type \"%s\" at the moment", type2char(Rtype));
	SEXP ans = PROTECT(colsum ? _new_Rmatrix0(Rtype, x_nrow, ng, R_NilValue)
				  : _new_Rmatrix0(Rtype, ng, x_ncol, R_NilValue));
	svt_view xv = make_view(x_dim, Rtype, x_SVT, 0);
	int ovflow = 0;
	int rc = colsum ? HIP_FN(svt_colsum_SVT)(&xv, INTEGER(group), ng, LOGICAL(na_rm)[0], DATAPTR(ans), &ovflow)
			: HIP_FN(svt_rowsum_SVT)(&xv, INTEGER(group), ng, LOGICAL(na_rm)[0], DATAPTR(ans), &ovflow);
	HIP_STATUS(rc, 1, colsum ? C_colsum_SVT_cpu(x_dim, x_type, x_SVT, group, ngroup, na_rm)
				 : C_rowsum_SVT_cpu(x_dim, x_type, x_SVT, group, ngroup, na_rm));
	if (ovflow)
		warning("NAs produced by integer overflow");
	UNPROTECT(1);
	return ans;
}

SEXP C_rowsum_SVT(SEXP x_dim, SEXP x_type, SEXP x_SVT, SEXP group, SEXP ngroup, SEXP na_rm)
{
	if (!hip_available())
		return C_rowsum_SVT_cpu(x_dim, x_type, x_SVT, group, ngroup, na_rm);
	return groupsum_SVT(x_dim, x_type, x_SVT, group, ngroup, na_rm, 0, "C_rowsum_SVT");
}

SEXP C_colsum_SVT(SEXP x_dim, SEXP x_type, SEXP x_SVT, SEXP group, SEXP ngroup, SEXP na_rm)
{
	if (!hip_available())
		return C_colsum_SVT_cpu(x_dim, x_type, x_SVT, group, ngroup, na_rm);
	return groupsum_SVT(x_dim, x_type, x_SVT, group, ngroup, na_rm, 1, "C_colsum_SVT");
}

static SEXP groupsum_dgCMatrix(SEXP x, SEXP group, SEXP ngroup, SEXP na_rm, int colsum)
{
	SEXP x_Dim = GET_SLOT(x, install("Dim"));
	int x_nrow = INTEGER(x_Dim)[0], x_ncol = INTEGER(x_Dim)[1];
	SEXP x_slotx = GET_SLOT(x, install("x")), x_sloti = GET_SLOT(x, install("i")),
	     x_slotp = GET_SLOT(x, install("p"));
	int ng = INTEGER(ngroup)[0];
	check_group(group, colsum ? x_ncol : x_nrow, ng);
	if ((double) ng * (double) (colsum ? x_nrow : x_ncol) > 2147483647.0)
		error("too many groups (matrix of sums will be too big)");
	SEXP ans = PROTECT(colsum ? _new_Rmatrix0(REALSXP, x_nrow, ng, R_NilValue)
				  : _new_Rmatrix0(REALSXP, ng, x_ncol, R_NilValue));
	int rc = colsum ? HIP_FN(svt_colsum_dgCMatrix)(x_nrow, x_ncol, REAL(x_slotx), INTEGER(x_sloti),
						       INTEGER(x_slotp), INTEGER(group), ng,
						       LOGICAL(na_rm)[0], REAL(ans))
			: HIP_FN(svt_rowsum_dgCMatrix)(x_nrow, x_ncol, REAL(x_slotx), INTEGER(x_sloti),
						       INTEGER(x_slotp), INTEGER(group), ng,
						       LOGICAL(na_rm)[0], REAL(ans));
	HIP_STATUS(rc, 1, colsum ? C_colsum_dgCMatrix_cpu(x, group, ngroup, na_rm)
				 : C_rowsum_dgCMatrix_cpu(x, group, ngroup, na_rm));
	UNPROTECT(1);
	return ans;
}

SEXP C_rowsum_dgCMatrix(SEXP x, SEXP group, SEXP ngroup, SEXP na_rm)
{
	if (!hip_available())
		return C_rowsum_dgCMatrix_cpu(x, group, ngroup, na_rm);
	return groupsum_dgCMatrix(x, group, ngroup, na_rm, 0);
}

SEXP C_colsum_dgCMatrix(SEXP x, SEXP group, SEXP ngroup, SEXP na_rm)
{
	if (!hip_available())
		return C_colsum_dgCMatrix_cpu(x, group, ngroup, na_rm);
	return groupsum_dgCMatrix(x, group, ngroup, na_rm, 1);
}

/* ------------------------------------------------------------------------------------------------ */
/* column statistics of a dgCMatrix -- src/sparseMatrix_utils.c:105-223                             */
/* ------------------------------------------------------------------------------------------------ */
typedef int (*dgc_colstat_fn)(int, int, const double *, const int *, int, double *);
typedef SEXP (*dgc_cpu_fn)(SEXP, SEXP);

static SEXP colstat_dgCMatrix(SEXP x, SEXP na_rm, dgc_colstat_fn fn, int is_range, dgc_cpu_fn cpu)
{
	SEXP x_Dim = GET_SLOT(x, install("Dim"));
	int x_nrow = INTEGER(x_Dim)[0], x_ncol = INTEGER(x_Dim)[1];
	SEXP x_slotx = GET_SLOT(x, install("x")), x_slotp = GET_SLOT(x, install("p"));
	SEXP ans = PROTECT(is_range ? allocMatrix(REALSXP, x_ncol, 2) : NEW_NUMERIC(x_ncol));
	HIP_STATUS(fn(x_nrow, x_ncol, REAL(x_slotx), INTEGER(x_slotp), LOGICAL(na_rm)[0], REAL(ans)), 1, cpu(x, na_rm));
	UNPROTECT(1);
	return ans;
}

SEXP C_colMins_dgCMatrix(SEXP x, SEXP na_rm)
{
	if (!hip_available())
		return C_colMins_dgCMatrix_cpu(x, na_rm);
	return colstat_dgCMatrix(x, na_rm, HIP_FN(svt_colMins_dgCMatrix), 0, C_colMins_dgCMatrix_cpu);
}

SEXP C_colMaxs_dgCMatrix(SEXP x, SEXP na_rm)
{
	if (!hip_available())
		return C_colMaxs_dgCMatrix_cpu(x, na_rm);
	return colstat_dgCMatrix(x, na_rm, HIP_FN(svt_colMaxs_dgCMatrix), 0, C_colMaxs_dgCMatrix_cpu);
}

SEXP C_colRanges_dgCMatrix(SEXP x, SEXP na_rm)
{
	if (!hip_available())
		return C_colRanges_dgCMatrix_cpu(x, na_rm);
	return colstat_dgCMatrix(x, na_rm, HIP_FN(svt_colRanges_dgCMatrix), 1, C_colRanges_dgCMatrix_cpu);
}

SEXP C_colVars_dgCMatrix(SEXP x, SEXP na_rm)
{
	if (!hip_available())
		return C_colVars_dgCMatrix_cpu(x, na_rm);
	return colstat_dgCMatrix(x, na_rm, HIP_FN(svt_colVars_dgCMatrix), 0, C_colVars_dgCMatrix_cpu);
}

/* ------------------------------------------------------------------------------------------------ */
/* t() / aperm() -- src/SparseArray_aperm.c:395-423, 1032-1056                                       */
/* The library hands back the CSC layout of the permuted array; the R tree is rebuilt from it with   */
/* the reference's own leaf constructor (_make_leaf_from_two_arrays, src/leaf_utils.c:100-128: a     */
/* leaf of ones comes out lacunar, an empty range as R_NilValue), NULL subtrees where nothing lands  */
/* (as REC_aperm_SVT leaves them, :1013-1029).                                                       */
/* ------------------------------------------------------------------------------------------------ */
/* leaves [first, first + prod(dim[1..ndim-1])) of the CSC triple -> the subtree over dim[0..ndim-1] */
static SEXP tree_from_csc(const int *dim, int ndim, SEXPTYPE Rtype, const int64_t *col_ptr,
			  const int *row_idx, const char *val, size_t esz, R_xlen_t first)
{
	if (ndim == 1) {
		int64_t a = col_ptr[first], n = col_ptr[first + 1] - a;
		return _make_leaf_from_two_arrays(Rtype, val + (size_t) a * esz, row_idx + a, (int) n);
	}
	R_xlen_t stride = 1;
	for (int k = 1; k < ndim - 1; k++)
		stride *= dim[k];
	if (col_ptr[first + stride * dim[ndim - 1]] == col_ptr[first])
		return R_NilValue;
	SEXP ans = PROTECT(NEW_LIST(dim[ndim - 1]));
	for (int i = 0; i < dim[ndim - 1]; i++) {
		SEXP elt = PROTECT(tree_from_csc(dim, ndim - 1, Rtype, col_ptr, row_idx, val, esz,
						 first + (R_xlen_t) i * stride));
		SET_VECTOR_ELT(ans, i, elt);
		UNPROTECT(1);
	}
	UNPROTECT(1);
	return ans;
}

static R_xlen_t view_nnz(const svt_view *v)
{
	R_xlen_t t = 0;
	for (int64_t j = 0; j < v->nleaves; j++)
		t += v->nzcount[j];
	return t;
}

SEXP C_transpose_2D_SVT(SEXP x_dim, SEXP x_type, SEXP x_SVT)
{
	SEXPTYPE Rtype = _get_and_check_Rtype_from_Rstring(x_type, "C_transpose_2D_SVT", "x_type");
	if (!hip_available() || !device_type(Rtype))         /* complex, raw, character, list: CPU */
		return C_transpose_2D_SVT_cpu(x_dim, x_type, x_SVT);
	if (LENGTH(x_dim) != 2)
		error("object to transpose must have exactly 2 dimensions");
	if (x_SVT == R_NilValue)
		return x_SVT;
	svt_view xv = make_view(x_dim, Rtype, x_SVT, 0);
	R_xlen_t nnz = view_nnz(&xv);
	size_t esz = Rtype == REALSXP ? 8 : 4;
	int ans_dim[2] = { INTEGER(x_dim)[1], INTEGER(x_dim)[0] };
	int64_t *cp = (int64_t *) R_alloc((size_t) ans_dim[1] + 1, sizeof(int64_t));
	int *ri = (int *) R_alloc(nnz > 0 ? (size_t) nnz : 1, sizeof(int));
	char *vv = (char *) R_alloc(nnz > 0 ? (size_t) nnz : 1, esz);
	HIP_STATUS(HIP_FN(svt_transpose_2D_SVT)(&xv, cp, ri, vv), 0, C_transpose_2D_SVT_cpu(x_dim, x_type, x_SVT));
	return tree_from_csc(ans_dim, 2, Rtype, cp, ri, vv, esz, 0);
}

SEXP C_aperm_SVT(SEXP x_dim, SEXP x_type, SEXP x_SVT, SEXP perm)
{
	SEXPTYPE Rtype = _get_and_check_Rtype_from_Rstring(x_type, "C_aperm_SVT", "x_type");
	int ndim = LENGTH(x_dim);
	if (!hip_available() || !device_type(Rtype) || ndim > 8)
		return C_aperm_SVT_cpu(x_dim, x_type, x_SVT, perm);
	/* check_perm(), src/SparseArray_aperm.c:455-474 */
	if (!IS_INTEGER(perm))
		error("'perm' must be an integer vector");
	if (LENGTH(perm) != ndim)
		error("'length(perm)' not equal to number of dimensions of array to permute");
	int identity = 1, taken[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	for (int a = 0; a < ndim; a++) {
		int q = INTEGER(perm)[a];
		if (q == NA_INTEGER || q < 1 || q > ndim)
			error("invalid 'perm' argument");
		if (taken[q - 1])
			error("'perm' cannot contain duplicates");
		taken[q - 1] = 1;
		if (q != a + 1)
			identity = 0;
	}
	if (identity || x_SVT == R_NilValue)                  /* :1044-1045 */
		return x_SVT;
	int *ans_dim = (int *) R_alloc(ndim, sizeof(int));
	R_xlen_t new_nl = 1;
	for (int a = 0; a < ndim; a++)
		ans_dim[a] = INTEGER(x_dim)[INTEGER(perm)[a] - 1];
	for (int a = 1; a < ndim; a++)
		new_nl *= ans_dim[a];
	svt_view xv = make_view(x_dim, Rtype, x_SVT, 0);
	R_xlen_t nnz = view_nnz(&xv);
	size_t esz = Rtype == REALSXP ? 8 : 4;
	int64_t *cp = (int64_t *) R_alloc((size_t) new_nl + 1, sizeof(int64_t));
	int *ri = (int *) R_alloc(nnz > 0 ? (size_t) nnz : 1, sizeof(int));
	char *vv = (char *) R_alloc(nnz > 0 ? (size_t) nnz : 1, esz);
	HIP_STATUS(HIP_FN(svt_aperm_SVT)(&xv, INTEGER(perm), cp, ri, vv), 0, C_aperm_SVT_cpu(x_dim, x_type, x_SVT, perm));
	return tree_from_csc(ans_dim, ndim, Rtype, cp, ri, vv, esz, 0);
}

/* ------------------------------------------------------------------------------------------------ */
/* thread control -- src/thread_control.c:47-66.  SparseArray.Call() (R/thread-control.R:87-92) sets */
/* the team size before every .Call and restores it afterwards: both paths must see the value.      */
/* ------------------------------------------------------------------------------------------------ */
SEXP C_get_num_procs(void)
{
	return C_get_num_procs_cpu();
}

SEXP C_get_max_threads(void)
{
	return C_get_max_threads_cpu();
}

SEXP C_set_max_threads(SEXP nthread)
{
	SEXP prev = C_set_max_threads_cpu(nthread);     /* the OpenMP team of the CPU bodies */
	if (hip_available())                           /* the library's host-side thread team (marshalling) */
		(void) HIP_FN(svt_set_max_threads)(INTEGER(nthread)[0]);
	return prev;
}
