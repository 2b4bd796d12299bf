/*
 * Test-only: the svt_* symbols integration/svt_hip_glue.c looks up with dlsym(), bound to the CPU
 * oracle's ABI (oracle/svt_oracle.h: same argument lists, same svt_view layout, prefix orc_), so that
 * tests/test_glue_executes.py can RUN the glue in the build container, where there is no GPU.  The
 * prototypes come from include/svt_hip.h: a mismatch between the two ABIs is a compile error here.
 * Never shipped, never loaded by the product (tests/test_abi.py forbids the oracle there).
 */
#include <stdlib.h>

#include "svt_hip.h"
#include "svt_oracle.h"

/* SVT_SHIM_STATUS=<n> in the environment: every compute entry point answers n without computing -- n = 1 is the
   library's "not supported here" (include/svt_hip.h), which the glue must answer with the reference's body */
static int forced(int *rc)
{
	const char *e = getenv("SVT_SHIM_STATUS");
	if (e == NULL || e[0] == '\0') return 0;
	*rc = atoi(e);
	return 1;
}
#define FORCED do { int rc__; if (forced(&rc__)) return rc__; } while (0)

int svt_init(int device) { (void) device; return 0; }
const char *svt_last_error(void) { int rc__; return forced(&rc__) ? "forced status" : orc_last_error(); }
int svt_set_max_threads(int n) { return orc_set_max_threads(n); }

#define V(x) ((const orc_svt *) (x))
int svt_crossprod2_SVT_mat(const svt_view *x, const void *y, int y_nrow, int y_ncol, int y_Rtype, int tr_y, double *out)
{ FORCED; return orc_crossprod2_SVT_mat(V(x), y, y_nrow, y_ncol, y_Rtype, tr_y, out); }
int svt_crossprod2_mat_SVT(const void *x, int x_nrow, int x_ncol, int x_Rtype, const svt_view *y, int tr_x, double *out)
{ FORCED; return orc_crossprod2_mat_SVT(x, x_nrow, x_ncol, x_Rtype, V(y), tr_x, out); }
int svt_crossprod2_SVT_SVT(const svt_view *x, const svt_view *y, double *out)
{ FORCED; return orc_crossprod2_SVT_SVT(V(x), V(y), out); }
int svt_crossprod1_SVT(const svt_view *x, double *out) { FORCED; return orc_crossprod1_SVT(V(x), out); }
int svt_summarize_SVT(const svt_view *x, int opcode, int na_rm, double center, double *out_d, int *out_i,
		      int *out_Rtype, int *warn)
{ FORCED; return orc_summarize_SVT(V(x), opcode, na_rm, center, out_d, out_i, out_Rtype, warn); }
int svt_colStats_out_Rtype(int opcode, int in_Rtype) { return orc_colStats_out_Rtype(opcode, in_Rtype); }
int svt_colStats_SVT(const svt_view *x, int opcode, int na_rm, double center, int dims, void *out, int *warn)
{ FORCED; return orc_colStats_SVT(V(x), opcode, na_rm, center, dims, out, warn); }
int svt_rowStats_SVT(const svt_view *x, int opcode, int na_rm, const double *center, int dims, void *out, int *warn)
{ FORCED; return orc_rowStats_SVT(V(x), opcode, na_rm, center, dims, out, warn); }
int svt_rowsum_SVT(const svt_view *x, const int *group, int ngroup, int na_rm, void *out, int *ovflow)
{ FORCED; return orc_rowsum_SVT(V(x), group, ngroup, na_rm, out, ovflow); }
int svt_colsum_SVT(const svt_view *x, const int *group, int ngroup, int na_rm, void *out, int *ovflow)
{ FORCED; return orc_colsum_SVT(V(x), group, ngroup, na_rm, out, ovflow); }
int svt_rowsum_dgCMatrix(int nrow, int ncol, const double *xx, const int *xi, const int *xp, const int *group,
			 int ngroup, int na_rm, double *out)
{ FORCED; return orc_rowsum_dgCMatrix(nrow, ncol, xx, xi, xp, group, ngroup, na_rm, out); }
int svt_colsum_dgCMatrix(int nrow, int ncol, const double *xx, const int *xi, const int *xp, const int *group,
			 int ngroup, int na_rm, double *out)
{ FORCED; return orc_colsum_dgCMatrix(nrow, ncol, xx, xi, xp, group, ngroup, na_rm, out); }
int svt_colMins_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp, int na_rm, double *out)
{ FORCED; return orc_colMins_dgCMatrix(nrow, ncol, xx, xp, na_rm, out); }
int svt_colMaxs_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp, int na_rm, double *out)
{ FORCED; return orc_colMaxs_dgCMatrix(nrow, ncol, xx, xp, na_rm, out); }
int svt_colRanges_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp, int na_rm, double *out)
{ FORCED; return orc_colRanges_dgCMatrix(nrow, ncol, xx, xp, na_rm, out); }
int svt_colVars_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp, int na_rm, double *out)
{ FORCED; return orc_colVars_dgCMatrix(nrow, ncol, xx, xp, na_rm, out); }
int svt_aperm_SVT(const svt_view *x, const int *perm, int64_t *out_col_ptr, int32_t *out_row_idx, void *out_val)
{ FORCED; return orc_aperm_SVT(V(x), perm, out_col_ptr, out_row_idx, out_val); }
int svt_transpose_2D_SVT(const svt_view *x, int64_t *out_col_ptr, int32_t *out_row_idx, void *out_val)
{ FORCED; return orc_transpose_2D_SVT(V(x), out_col_ptr, out_row_idx, out_val); }

/* the layouts must agree field by field */
_Static_assert(sizeof(svt_view) == sizeof(orc_svt), "svt_view and orc_svt differ in size");
