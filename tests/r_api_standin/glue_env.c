/*
 * Test-only environment of integration/svt_hip_glue.c for tests/test_glue_executes.py: what the glue
 * expects to find in the R package it is added to, beyond the helper files that compile as they are
 * (src/argcheck_utils.c, src/Rvector_utils.c, src/Rvector_summarization.c are compiled from the
 * read-only mount by the test):
 *
 *  - the helpers that are `static` in the reference and that its maintainer makes extern for the glue
 *    (list in the glue's header comment).  A static function cannot be linked from where it lies, so
 *    the test supplies its own small versions with the documented behaviour (result geometry =
 *    tail / head of dim, names for 1-d results, dimnames dropped when every retained entry is NULL;
 *    src/SparseArray_matrixStats.c:33-176, 1079-1097; src/rowsum_methods.c:15-37);
 *  - the leaf constructor _make_leaf_from_two_arrays (src/leaf_utils.c:100-128 needs the S4Vectors
 *    headers, absent from the image): empty range -> R_NilValue, all ones -> lacunar leaf
 *    list(NULL, nzoffs), else list(nzvals, nzoffs);
 *  - the reference's bodies under their `_cpu` names: the test runs with the library "available", so a
 *    call that lands in one of them is recorded (and fails the test) -- except the thread-control trio,
 *    which the glue always forwards to, bound here to the oracle's team size.
 *
 * Nothing here is shipped, linked into the product or used as an oracle.
 */
#include <Rdefines.h>
#include <string.h>

#include "leaf_utils.h"

int orc_get_num_procs(void);
int orc_get_max_threads(void);
int orc_set_max_threads(int);

/* ---- helpers the maintainer makes extern ----------------------------------------------------- */
int check_dims(SEXP dims, int min, int max)
{
	int d = IS_INTEGER(dims) && LENGTH(dims) == 1 ? INTEGER(dims)[0] : NA_INTEGER;
	if (!IS_INTEGER(dims) || LENGTH(dims) != 1)
		error("'dims' must be a single integer");
	if (d == NA_INTEGER || d < min || d > max)
		error("'dims' must be >= %d and <= %d", min, max);
	return d;
}

static SEXP dim_slice(SEXP x_dim, int from, int n)
{
	SEXP out = PROTECT(NEW_INTEGER(n));
	for (int a = 0; a < n; a++)
		INTEGER(out)[a] = INTEGER(x_dim)[from + a];
	UNPROTECT(1);
	return out;
}

SEXP compute_colStats_ans_dim(SEXP x_dim, int dims)
{
	return dim_slice(x_dim, dims, LENGTH(x_dim) - dims);
}

SEXP compute_rowStats_ans_dim(SEXP x_dim, int ans_ndim)
{
	return dim_slice(x_dim, 0, ans_ndim);
}

SEXP alloc_ans(SEXPTYPE Rtype, SEXP ans_dim, R_xlen_t *out_incs)
{
	int nd = LENGTH(ans_dim);
	SEXP ans = PROTECT(nd >= 2 ? allocArray(Rtype, ans_dim)
				   : allocVector(Rtype, nd == 1 ? INTEGER(ans_dim)[0] : 1));
	R_xlen_t inc = 1;
	for (int a = 0; a < nd; a++) {
		out_incs[a] = inc;
		inc *= INTEGER(ans_dim)[a];
	}
	UNPROTECT(1);
	return ans;
}

/* entries [from, from + n) of x_dimnames -> names (n == 1) or dimnames (n >= 2) of ans */
static void carry_dimnames(SEXP ans, SEXP x_dimnames, int from, int n)
{
	if (x_dimnames == R_NilValue || n == 0)
		return;
	if (n == 1) {
		if (VECTOR_ELT(x_dimnames, from) != R_NilValue)
			SET_NAMES(ans, VECTOR_ELT(x_dimnames, from));
		return;
	}
	int any = 0;
	for (int a = 0; a < n; a++)
		any |= VECTOR_ELT(x_dimnames, from + a) != R_NilValue;
	if (!any)
		return;
	SEXP dn = PROTECT(NEW_LIST(n));
	for (int a = 0; a < n; a++)
		SET_VECTOR_ELT(dn, a, VECTOR_ELT(x_dimnames, from + a));
	SET_DIMNAMES(ans, dn);
	UNPROTECT(1);
}

void propagate_colStats_dimnames(SEXP ans, SEXP x_dimnames, int dims)
{
	if (x_dimnames != R_NilValue)
		carry_dimnames(ans, x_dimnames, dims, LENGTH(x_dimnames) - dims);
}

void propagate_rowStats_dimnames(SEXP ans, SEXP x_dimnames, int dims)
{
	carry_dimnames(ans, x_dimnames, 0, dims);
}

const double *check_rowStats_center(SEXP center, SEXP x_dim, int ans_ndim)
{
	if (center == R_NilValue)
		return NULL;
	if (!IS_NUMERIC(center))
		error("SparseArray internal error in check_rowStats_center():\n"
		      "    'center' must be NULL or a numeric array");
	R_xlen_t want = 1;
	for (int a = 0; a < ans_ndim; a++)
		want *= INTEGER(x_dim)[a];
	if (LENGTH(center) != want)
		error("SparseArray internal error in check_rowStats_center():\n"
		      "    unexpected 'center' length");
	return REAL(center);
}

void check_group(SEXP group, int x_nrow, int ngroup)
{
	if (!IS_INTEGER(group))
		error("the grouping vector must be an integer vector or factor");
	if (LENGTH(group) != x_nrow)
		error("the grouping vector must have one element per row in 'x' for rowsum()\n"
		      "  and one element per column in 'x' for colsum()");
	for (int i = 0; i < x_nrow; i++) {
		int g = INTEGER(group)[i];
		if (g == NA_INTEGER ? ngroup < 1 : (g < 1 || g > ngroup))
			error(g == NA_INTEGER ? "'ngroup' must be >= 1 when 'group' contains missing values"
					      : "all non-NA values in 'group' must be >= 1 and <= 'ngroup'");
	}
}

/* ---- the leaf constructor ---------------------------------------------------------------------- */
SEXP _make_leaf_from_two_arrays(SEXPTYPE Rtype, const void *nzvals_p, const int *nzoffs_p, int nzcount)
{
	if (nzcount == 0)
		return R_NilValue;
	size_t esz = Rtype == REALSXP ? 8 : 4;
	int ones = 1;
	for (int k = 0; k < nzcount && ones; k++)
		ones = Rtype == REALSXP ? ((const double *) nzvals_p)[k] == 1.0 : ((const int *) nzvals_p)[k] == 1;
	SEXP offs = PROTECT(NEW_INTEGER(nzcount));
	memcpy(INTEGER(offs), nzoffs_p, sizeof(int) * (size_t) nzcount);
	SEXP vals = R_NilValue;
	if (!ones) {
		vals = allocVector(Rtype, nzcount);
		memcpy(DATAPTR(vals), nzvals_p, esz * (size_t) nzcount);
	}
	PROTECT(vals);
	SEXP leaf = PROTECT(NEW_LIST(2));
	SET_VECTOR_ELT(leaf, 0, vals);
	SET_VECTOR_ELT(leaf, 1, offs);
	UNPROTECT(3);
	return leaf;
}

/* ---- the reference's bodies under their _cpu names --------------------------------------------- */
static int cpu_body_calls;
static char cpu_body_last[64];
int env_cpu_body_calls(void) { return cpu_body_calls; }
const char *env_cpu_body_last(void) { return cpu_body_last; }
void env_cpu_body_reset(void) { cpu_body_calls = 0; cpu_body_last[0] = 0; }

#define CPU_BODY(name, ...) \
	SEXP name(__VA_ARGS__) { cpu_body_calls++; strncpy(cpu_body_last, #name, sizeof(cpu_body_last) - 1); return R_NilValue; }
#define S SEXP
CPU_BODY(C_crossprod2_SVT_mat_cpu, S a, S b, S c, S d, S e, S f, S g)
CPU_BODY(C_crossprod2_mat_SVT_cpu, S a, S b, S c, S d, S e, S f, S g)
CPU_BODY(C_crossprod2_SVT_SVT_cpu, S a, S b, S c, S d, S e, S f, S g, S h)
CPU_BODY(C_crossprod1_SVT_cpu, S a, S b, S c, S d, S e)
CPU_BODY(C_colStats_SVT_cpu, S a, S b, S c, S d, S e, S f, S g, S h, S i)
CPU_BODY(C_rowStats_SVT_cpu, S a, S b, S c, S d, S e, S f, S g, S h, S i)
CPU_BODY(C_summarize_SVT_cpu, S a, S b, S c, S d, S e, S f, S g)
CPU_BODY(C_rowsum_SVT_cpu, S a, S b, S c, S d, S e, S f)
CPU_BODY(C_colsum_SVT_cpu, S a, S b, S c, S d, S e, S f)
CPU_BODY(C_rowsum_dgCMatrix_cpu, S a, S b, S c, S d)
CPU_BODY(C_colsum_dgCMatrix_cpu, S a, S b, S c, S d)
CPU_BODY(C_transpose_2D_SVT_cpu, S a, S b, S c)
CPU_BODY(C_aperm_SVT_cpu, S a, S b, S c, S d)
CPU_BODY(C_colMins_dgCMatrix_cpu, S a, S b)
CPU_BODY(C_colMaxs_dgCMatrix_cpu, S a, S b)
CPU_BODY(C_colRanges_dgCMatrix_cpu, S a, S b)
CPU_BODY(C_colVars_dgCMatrix_cpu, S a, S b)
#undef S

/* src/thread_control.c:47-66, on the oracle's OpenMP team */
SEXP C_get_num_procs_cpu(void) { return ScalarInteger(orc_get_num_procs()); }
SEXP C_get_max_threads_cpu(void) { return ScalarInteger(orc_get_max_threads()); }
SEXP C_set_max_threads_cpu(SEXP nthread) { return ScalarInteger(orc_set_max_threads(INTEGER(nthread)[0])); }
