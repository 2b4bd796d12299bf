/*
 * svt_oracle.c -- TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 *
 * Plain-C restatement of the algorithms on SparseArray's SVT compute hot
 * path.  Loop nests, summation order and NA/NaN rules follow the reference
 * (file:line given at each function, relative to the reference's src/).
 * The OpenMP pragmas sit at the same loops as the reference's
 * (SparseMatrix_mult.c:131-296, SparseArray_matrixStats.c:219).
 */
#include "svt_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * R's missing values.  NA_real_ is the quiet NaN whose low word is 1954
 * (R's arithmetic.c); NA_integer_ is INT_MIN.
 */
#define NA_INT INT_MIN

static double make_NA_real(void)
{
	union { double d; uint64_t u; } x;
	x.u = 0x7FF00000000007A2ULL;
	return x.d;
}
static inline int is_R_NA(double x)
{
	union { double d; uint64_t u; } y;
	if (!isnan(x))
		return 0;
	y.d = x;
	return (uint32_t) (y.u & 0xFFFFFFFFu) == 1954u;
}
static inline int is_R_NaN(double x)   /* NaN but not NA */
{
	return isnan(x) && !is_R_NA(x);
}
#define NA_REAL (make_NA_real())

static __thread char errbuf[512];

static int fail(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(errbuf, sizeof(errbuf), fmt, ap);
	va_end(ap);
	return -1;
}

const char *orc_last_error(void) { return errbuf; }

/* thread_control.c:12-64 */
int orc_get_num_procs(void)
{
#ifdef _OPENMP
	return omp_get_num_procs();
#else
	return 0;
#endif
}
int orc_get_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 0;
#endif
}
int orc_set_max_threads(int n)
{
#ifdef _OPENMP
	int prev = omp_get_max_threads();
	omp_set_num_threads(n);
	return prev;
#else
	(void) n;
	return 0;
#endif
}

/* A leaf as the kernels see it (SparseVec.h:11-18, na_background == 0). */
typedef struct {
	const int *off;
	const void *val;   /* NULL: lacunar, all ones */
	int n;
	int len;
} leaf_t;

static inline leaf_t get_leaf(const orc_svt *x, int64_t j)
{
	leaf_t lf;
	lf.n = x->svt_is_null ? 0 : x->nzcount[j];
	lf.off = lf.n ? x->nzoffs[j] : NULL;
	lf.val = lf.n ? x->nzvals[j] : NULL;
	lf.len = x->dim[0];
	return lf;
}

/* ========================================================================
 * Dot products -- SparseVec_dotprod.c
 */

/* SparseVec_dotprod.c:9-20 with the merge iterator of SparseVec.h:166-194,
   273-312: union of the two offset sets in ascending order, the absent side
   contributes 0, a lacunar side contributes 1. */
double orc_dotprod_doubleSV_doubleSV(const int *off1, const double *v1, int n1,
				     const int *off2, const double *v2, int n2)
{
	double acc = 0.0;
	int k1 = 0, k2 = 0;
	while (k1 < n1 || k2 < n2) {
		double a, b;
		int take1, take2;
		if (k1 < n1 && k2 < n2) {
			take1 = off1[k1] <= off2[k2];
			take2 = off2[k2] <= off1[k1];
		} else {
			take1 = k1 < n1;
			take2 = !take1;
		}
		a = take1 ? (v1 ? v1[k1] : 1.0) : 0.0;
		b = take2 ? (v2 ? v2[k2] : 1.0) : 0.0;
		k1 += take1;
		k2 += take2;
		if (is_R_NA(a) || is_R_NA(b))
			return NA_REAL;
		acc += a * b;
	}
	return acc;
}

/* SparseVec_dotprod.c:28-43 */
double orc_dotprod_doubleSV_finite_doubles(const int *off1, const double *v1,
					   int n1, const double *x2)
{
	double acc = 0.0;
	if (v1 == NULL) {
		for (int k = 0; k < n1; k++)
			acc += x2[off1[k]];
	} else {
		for (int k = 0; k < n1; k++)
			acc += v1[k] * x2[off1[k]];
	}
	return acc;
}

/* SparseVec_dotprod.c:48-65 -- visits every row, zeros included */
double orc_dotprod_doubleSV_doubles(const int *off1, const double *v1, int n1,
				    int len, const double *x2)
{
	double acc = 0.0;
	int k = 0;
	for (int i = 0; i < len; i++) {
		double a = 0.0, b = x2[i];
		if (is_R_NA(b))
			return NA_REAL;
		if (k < n1 && off1[k] == i) {
			a = v1 ? v1[k] : 1.0;
			if (is_R_NA(a))
				return NA_REAL;
			k++;
		}
		acc += a * b;
	}
	return acc;
}

/* SparseVec_dotprod.c:73-92 */
double orc_dotprod_intSV_noNA_ints(const int *off1, const int *v1, int n1,
				   const int *x2)
{
	double acc = 0.0;
	if (v1 == NULL) {
		for (int k = 0; k < n1; k++)
			acc += (double) x2[off1[k]];
		return acc;
	}
	for (int k = 0; k < n1; k++) {
		int a = v1[k];
		if (a == NA_INT)
			return NA_REAL;
		acc += (double) a * x2[off1[k]];
	}
	return acc;
}

/* SparseVec_dotprod.c:97-114 */
double orc_dotprod_intSV_ints(const int *off1, const int *v1, int n1,
			      int len, const int *x2)
{
	double acc = 0.0;
	int k = 0;
	for (int i = 0; i < len; i++) {
		int a = 0, b = x2[i];
		if (b == NA_INT)
			return NA_REAL;
		if (k < n1 && off1[k] == i) {
			a = v1 ? v1[k] : 1;
			if (a == NA_INT)
				return NA_REAL;
			k++;
		}
		acc += (double) a * b;
	}
	return acc;
}

/* SparseVec_dotprod.c:116-126 */
double orc_dotprod_doubles_zero(const double *x, int n)
{
	double acc = 0.0;
	for (int i = 0; i < n; i++) {
		if (is_R_NA(x[i]))
			return NA_REAL;
		acc += x[i] * 0.0;
	}
	return acc;
}

/* SparseVec_dotprod.c:128-138 */
double orc_dotprod_ints_zero(const int *x, int n)
{
	double acc = 0.0;
	for (int i = 0; i < n; i++) {
		if (x[i] == NA_INT)
			return NA_REAL;
		acc += (double) x[i] * 0.0;
	}
	return acc;
}

/* SparseVec_dotprod.c:140-156 */
static double dotprod_doubleSV_zero(const leaf_t *lf)
{
	if (lf->val == NULL)
		return 0.0;
	return orc_dotprod_doubles_zero((const double *) lf->val, lf->n);
}

/* ========================================================================
 * crossprod -- SparseMatrix_mult.c
 */

/* SparseMatrix_mult.c:23-54 */
static int all_finite(const double *x, int n)
{
	for (int i = 0; i < n; i++)
		if (!isfinite(x[i]))
			return 0;
	return 1;
}
static int no_int_NA(const int *x, int n)
{
	for (int i = 0; i < n; i++)
		if (x[i] == NA_INT)
			return 0;
	return 1;
}
static int leaf_all_finite(const leaf_t *lf)
{
	return lf->val == NULL ||
	       all_finite((const double *) lf->val, lf->n);
}
static int leaf_no_int_NA(const leaf_t *lf)
{
	return lf->val == NULL || no_int_NA((const int *) lf->val, lf->n);
}

/* The 5 per-leaf wrappers, SparseMatrix_mult.c:80-120.  An empty leaf is
   special-cased before the SparseVec is even built. */
static double leaf_dot_finite_doubles(const leaf_t *lf, const double *x2)
{
	if (lf->n == 0)
		return 0.0;
	return orc_dotprod_doubleSV_finite_doubles(lf->off,
			(const double *) lf->val, lf->n, x2);
}
static double leaf_dot_noNA_ints(const leaf_t *lf, const int *x2)
{
	if (lf->n == 0)
		return 0.0;
	return orc_dotprod_intSV_noNA_ints(lf->off, (const int *) lf->val,
					   lf->n, x2);
}
static double leaf_dot_doubles(const leaf_t *lf, const double *x2, int len)
{
	if (lf->n == 0)
		return orc_dotprod_doubles_zero(x2, len);
	return orc_dotprod_doubleSV_doubles(lf->off, (const double *) lf->val,
					    lf->n, len, x2);
}
static double leaf_dot_ints(const leaf_t *lf, const int *x2, int len)
{
	if (lf->n == 0)
		return orc_dotprod_ints_zero(x2, len);
	return orc_dotprod_intSV_ints(lf->off, (const int *) lf->val,
				      lf->n, len, x2);
}
static double leaf_dot_doubleSV(const leaf_t *lf, const leaf_t *sv2)
{
	if (lf->n == 0)
		return dotprod_doubleSV_zero(sv2);
	return orc_dotprod_doubleSV_doubleSV(lf->off,
			(const double *) lf->val, lf->n,
			sv2->off, (const double *) sv2->val, sv2->n);
}

/*
 * One dense vector against every leaf of an SVT.  'stride' is the distance
 * between consecutive results: 1 writes a column of 'out' (the *_Rcol
 * family, SparseMatrix_mult.c:143-152,166-175,193-207,225-239), out_nrow
 * writes a row (the *_Lcol family, :131-141,154-164,177-191,209-223).
 * 'prescan' says whether the caller is one of the "double"/"int" variants
 * that test the dense vector first, or already knows it is clean.
 */
static void dense_vs_leaves_double(const orc_svt *svt, const double *vec,
				   int len, int known_finite,
				   double *out, int64_t stride)
{
	int64_t n = svt->nleaves;
	if (known_finite || all_finite(vec, len)) {
		#pragma omp parallel for schedule(static)
		for (int64_t j = 0; j < n; j++) {
			leaf_t lf = get_leaf(svt, j);
			out[j * stride] = leaf_dot_finite_doubles(&lf, vec);
		}
		return;
	}
	#pragma omp parallel for schedule(static)
	for (int64_t j = 0; j < n; j++) {
		leaf_t lf = get_leaf(svt, j);
		out[j * stride] = leaf_dot_doubles(&lf, vec, len);
	}
}

static void dense_vs_leaves_int(const orc_svt *svt, const int *vec,
				int len, int known_clean,
				double *out, int64_t stride)
{
	int64_t n = svt->nleaves;
	if (known_clean || no_int_NA(vec, len)) {
		#pragma omp parallel for schedule(static)
		for (int64_t j = 0; j < n; j++) {
			leaf_t lf = get_leaf(svt, j);
			out[j * stride] = leaf_dot_noNA_ints(&lf, vec);
		}
		return;
	}
	#pragma omp parallel for schedule(static)
	for (int64_t j = 0; j < n; j++) {
		leaf_t lf = get_leaf(svt, j);
		out[j * stride] = leaf_dot_ints(&lf, vec, len);
	}
}

/* SparseMatrix_mult.c:241-261 */
static void sparse_vs_leaves_double(const orc_svt *svt, const leaf_t *sv,
				    double *out, int64_t stride)
{
	int64_t n = svt->nleaves;
	#pragma omp parallel for schedule(static)
	for (int64_t j = 0; j < n; j++) {
		leaf_t lf = get_leaf(svt, j);
		out[j * stride] = leaf_dot_doubleSV(&lf, sv);
	}
}

static int check_2d(const orc_svt *x, const char *what)
{
	if (x->ndim != 2)
		return fail("%s must have 2 dimensions", what);
	if (x->Rtype != ORC_DBL && x->Rtype != ORC_INT)
		return fail("input type is not supported yet");
	return 0;
}

/* SparseMatrix_mult.c:385-431 (double), :484-515 (int) and the entry point
   :931-982.  Dense columns outermost, leaves inside. */
int orc_crossprod2_SVT_mat(const orc_svt *x, const void *y, int y_nrow,
			   int y_ncol, int y_Rtype, int tr_y, double *out)
{
	if (check_2d(x, "input objects"))
		return -1;
	int in_nrow = x->dim[0], out_nrow = x->dim[1];
	if (in_nrow != (tr_y ? y_ncol : y_nrow))
		return fail("input objects are non-conformable");
	if (x->Rtype != y_Rtype)
		return fail("'x_Rtype != TYPEOF(y)' not supported yet");
	int out_ncol = tr_y ? y_nrow : y_ncol;
	memset(out, 0, sizeof(double) * (size_t) out_nrow * out_ncol);
	if (x->svt_is_null)
		return 0;
	size_t esz = x->Rtype == ORC_DBL ? sizeof(double) : sizeof(int);
	void *colbuf = tr_y ? malloc(esz * (in_nrow ? in_nrow : 1)) : NULL;
	for (int j = 0; j < out_ncol; j++) {
		double *outcol = out + (size_t) j * out_nrow;
		if (x->Rtype == ORC_DBL) {
			const double *col;
			if (tr_y) {   /* gather row j of y, :411-421 */
				double *b = (double *) colbuf;
				const double *src = (const double *) y + j;
				for (int i = 0; i < in_nrow; i++)
					b[i] = src[(size_t) i * out_ncol];
				col = b;
			} else {
				col = (const double *) y +
				      (size_t) j * in_nrow;
			}
			dense_vs_leaves_double(x, col, in_nrow, 0, outcol, 1);
		} else {
			const int *col;
			if (tr_y) {
				int *b = (int *) colbuf;
				const int *src = (const int *) y + j;
				for (int i = 0; i < in_nrow; i++)
					b[i] = src[(size_t) i * out_ncol];
				col = b;
			} else {
				col = (const int *) y + (size_t) j * in_nrow;
			}
			dense_vs_leaves_int(x, col, in_nrow, 0, outcol, 1);
		}
	}
	free(colbuf);
	return 0;
}

/* SparseMatrix_mult.c:435-479 (double), :519-547 (int), entry :985-1034 */
int orc_crossprod2_mat_SVT(const void *x, int x_nrow, int x_ncol, int x_Rtype,
			   const orc_svt *y, int tr_x, double *out)
{
	if (check_2d(y, "input objects"))
		return -1;
	int in_nrow = y->dim[0], out_ncol = y->dim[1];
	if ((tr_x ? x_ncol : x_nrow) != in_nrow)
		return fail("input objects are non-conformable");
	if (x_Rtype != y->Rtype)
		return fail("input objects must have the same type() for now");
	int out_nrow = tr_x ? x_nrow : x_ncol;
	memset(out, 0, sizeof(double) * (size_t) out_nrow * out_ncol);
	if (y->svt_is_null)
		return 0;
	size_t esz = y->Rtype == ORC_DBL ? sizeof(double) : sizeof(int);
	void *colbuf = tr_x ? malloc(esz * (in_nrow ? in_nrow : 1)) : NULL;
	for (int i = 0; i < out_nrow; i++) {
		double *outrow = out + i;
		if (y->Rtype == ORC_DBL) {
			const double *col;
			if (tr_x) {
				double *b = (double *) colbuf;
				const double *src = (const double *) x + i;
				for (int k = 0; k < in_nrow; k++)
					b[k] = src[(size_t) k * out_nrow];
				col = b;
			} else {
				col = (const double *) x +
				      (size_t) i * in_nrow;
			}
			dense_vs_leaves_double(y, col, in_nrow, 0,
					       outrow, out_nrow);
		} else {
			const int *col;
			if (tr_x) {
				int *b = (int *) colbuf;
				const int *src = (const int *) x + i;
				for (int k = 0; k < in_nrow; k++)
					b[k] = src[(size_t) k * out_nrow];
				col = b;
			} else {
				col = (const int *) x + (size_t) i * in_nrow;
			}
			dense_vs_leaves_int(y, col, in_nrow, 0,
					    outrow, out_nrow);
		}
	}
	free(colbuf);
	return 0;
}

/* SparseVec.c:9-47 (zero background) */
static void expand_leaf_double(const leaf_t *lf, double *dense)
{
	memset(dense, 0, sizeof(double) * lf->len);
	const double *v = (const double *) lf->val;
	for (int k = 0; k < lf->n; k++)
		dense[lf->off[k]] = v ? v[k] : 1.0;
}
static void expand_leaf_int(const leaf_t *lf, int *dense)
{
	memset(dense, 0, sizeof(int) * lf->len);
	const int *v = (const int *) lf->val;
	for (int k = 0; k < lf->n; k++)
		dense[lf->off[k]] = v ? v[k] : 1;
}

static int64_t total_nzcount(const orc_svt *x)   /* SVT_SparseArray_class.c:200-218 */
{
	int64_t tot = 0;
	if (x->svt_is_null)
		return 0;
	for (int64_t j = 0; j < x->nleaves; j++)
		tot += x->nzcount[j];
	return tot;
}

/*
 * One "preprocessed" leaf (expanded to dense when clean) against all the
 * leaves of the other operand: SparseMatrix_mult.c:632-724.
 * 'stride' as in dense_vs_leaves_*; 'nout' = number of results.
 */
static void pp_leaf_vs_leaves_double(const leaf_t *pp, const orc_svt *other,
				     double *densebuf, double *out,
				     int64_t stride)
{
	if (pp->n == 0) {
		memset(densebuf, 0, sizeof(double) * pp->len);
		dense_vs_leaves_double(other, densebuf, pp->len, 1,
				       out, stride);
		return;
	}
	if (leaf_all_finite(pp)) {
		expand_leaf_double(pp, densebuf);
		dense_vs_leaves_double(other, densebuf, pp->len, 1,
				       out, stride);
		return;
	}
	sparse_vs_leaves_double(other, pp, out, stride);
}

static void pp_leaf_vs_leaves_int(const leaf_t *pp, const orc_svt *other,
				  int *densebuf, double *out, int64_t stride)
{
	if (pp->n == 0) {
		memset(densebuf, 0, sizeof(int) * pp->len);
		dense_vs_leaves_int(other, densebuf, pp->len, 1, out, stride);
		return;
	}
	if (leaf_no_int_NA(pp)) {
		expand_leaf_int(pp, densebuf);
		dense_vs_leaves_int(other, densebuf, pp->len, 1, out, stride);
		return;
	}
	/* fill_row / fill_col with NA, :690,:722 */
	for (int64_t j = 0; j < other->nleaves; j++)
		out[j * stride] = NA_REAL;
}

/* One operand is the all-zero matrix: SparseMatrix_mult.c:558-628.
   'fill_stride'/'fill_n' describe the row or column of 'out' that belongs
   to leaf j of the non-NULL operand. */
static void crossprod2_with_mat0(const orc_svt *svt, double *out,
				 int64_t leaf_stride, int64_t fill_stride,
				 int64_t fill_n)
{
	if (svt->svt_is_null)
		return;
	for (int64_t j = 0; j < svt->nleaves; j++) {
		leaf_t lf = get_leaf(svt, j);
		if (lf.n == 0)
			continue;
		double v;
		if (svt->Rtype == ORC_DBL) {
			v = dotprod_doubleSV_zero(&lf);
		} else {
			if (leaf_no_int_NA(&lf))
				continue;
			v = NA_REAL;
		}
		double *p = out + j * leaf_stride;
		for (int64_t i = 0; i < fill_n; i++)
			p[i * fill_stride] = v;
	}
}

/* SparseMatrix_mult.c:1037-1101 with :728-820 */
int orc_crossprod2_SVT_SVT(const orc_svt *x, const orc_svt *y, double *out)
{
	if (check_2d(x, "input objects") || check_2d(y, "input objects"))
		return -1;
	int in_nrow = x->dim[0];
	if (in_nrow != y->dim[0])
		return fail("input SVT_SparseMatrix objects "
			    "are non-conformable");
	if (x->Rtype != y->Rtype)
		return fail("input SVT_SparseMatrix objects "
			    "must have the same type() for now");
	int out_nrow = x->dim[1], out_ncol = y->dim[1];
	memset(out, 0, sizeof(double) * (size_t) out_nrow * out_ncol);

	int64_t Lpp_nops = total_nzcount(y) * out_nrow;
	int64_t Rpp_nops = total_nzcount(x) * out_ncol;
	size_t esz = x->Rtype == ORC_DBL ? sizeof(double) : sizeof(int);
	if (Lpp_nops < Rpp_nops) {
		/* preprocess the leaves of x, fill 'out' row by row */
		if (y->svt_is_null) {
			crossprod2_with_mat0(x, out, 1, out_nrow, out_ncol);
			return 0;
		}
		void *densebuf = malloc(esz * (in_nrow ? in_nrow : 1));
		for (int i = 0; i < out_nrow; i++) {
			leaf_t pp = get_leaf(x, i);
			if (x->Rtype == ORC_DBL)
				pp_leaf_vs_leaves_double(&pp, y,
					(double *) densebuf, out + i, out_nrow);
			else
				pp_leaf_vs_leaves_int(&pp, y,
					(int *) densebuf, out + i, out_nrow);
		}
		free(densebuf);
	} else {
		/* preprocess the leaves of y, fill 'out' column by column */
		if (x->svt_is_null) {
			crossprod2_with_mat0(y, out, out_nrow, 1, out_nrow);
			return 0;
		}
		void *densebuf = malloc(esz * (in_nrow ? in_nrow : 1));
		for (int j = 0; j < out_ncol; j++) {
			leaf_t pp = get_leaf(y, j);
			double *outcol = out + (size_t) j * out_nrow;
			if (x->Rtype == ORC_DBL)
				pp_leaf_vs_leaves_double(&pp, x,
					(double *) densebuf, outcol, 1);
			else
				pp_leaf_vs_leaves_int(&pp, x,
					(int *) densebuf, outcol, 1);
		}
		free(densebuf);
	}
	return 0;
}

/* SparseMatrix_mult.c:1104-1140 with :263-296, :827-908.
   Column j: the diagonal cell, then cells (j+k, j) and (j, j+k), k >= 1. */
int orc_crossprod1_SVT(const orc_svt *x, double *out)
{
	if (check_2d(x, "'x'"))
		return -1;
	int in_nrow = x->dim[0], n = x->dim[1];
	memset(out, 0, sizeof(double) * (size_t) n * n);
	if (x->svt_is_null)
		return 0;
	size_t esz = x->Rtype == ORC_DBL ? sizeof(double) : sizeof(int);
	void *densebuf = malloc(esz * (in_nrow ? in_nrow : 1));
	for (int j = 0; j < n; j++) {
		double *diag = out + (size_t) j * n + j;
		leaf_t lf = get_leaf(x, j);
		int mode;   /* 0: dense col, 1: sparse merge, 2: NA fill */
		if (x->Rtype == ORC_DBL) {
			double *dense = (double *) densebuf;
			if (lf.n == 0) {
				memset(dense, 0, sizeof(double) * in_nrow);
				mode = 0;
			} else if (leaf_all_finite(&lf)) {
				expand_leaf_double(&lf, dense);
				*diag = orc_dotprod_doubleSV_finite_doubles(
					lf.off, (const double *) lf.val,
					lf.n, dense);
				mode = 0;
			} else {
				*diag = orc_dotprod_doubleSV_doubleSV(
					lf.off, (const double *) lf.val, lf.n,
					lf.off, (const double *) lf.val, lf.n);
				mode = 1;
			}
			#pragma omp parallel for schedule(static)
			for (int k = n - 1 - j; k >= 1; k--) {
				leaf_t other = get_leaf(x, j + k);
				double dp = mode == 0 ?
					leaf_dot_finite_doubles(&other, dense) :
					leaf_dot_doubleSV(&other, &lf);
				diag[k] = diag[(size_t) k * n] = dp;
			}
		} else {
			int *dense = (int *) densebuf;
			if (lf.n == 0) {
				memset(dense, 0, sizeof(int) * in_nrow);
				mode = 0;
			} else if (leaf_no_int_NA(&lf)) {
				expand_leaf_int(&lf, dense);
				*diag = orc_dotprod_intSV_noNA_ints(lf.off,
					(const int *) lf.val, lf.n, dense);
				mode = 0;
			} else {
				mode = 2;
			}
			if (mode == 2) {   /* sym_fill_with_NAs, :70-78 */
				*diag = NA_REAL;
				for (int k = 1; k < n - j; k++)
					diag[k] = diag[(size_t) k * n] = NA_REAL;
				continue;
			}
			#pragma omp parallel for schedule(static)
			for (int k = n - 1 - j; k >= 1; k--) {
				leaf_t other = get_leaf(x, j + k);
				diag[k] = diag[(size_t) k * n] =
					leaf_dot_noNA_ints(&other, dense);
			}
		}
	}
	free(densebuf);
	return 0;
}

/* ========================================================================
 * Summarization of one vector of values -- Rvector_summarization.c
 */

enum { ST_NOT_SET = 1, ST_SET = 2, ST_BREAK = 3 };   /* .h:50-52 */

typedef struct {
	int opcode, in_Rtype, na_rm;
	double center;
	int na_bg;        /* NaArray: the implicit value is NA (Rvector_summarization.c:1078-1106) */
} sum_op;

typedef struct {
	int64_t in_length, in_nzcount, in_nacount;
	int out_Rtype;
	int status;
	union { int i[2]; double d[2]; } buf;
	int one_zero;     /* implicit zeros must be fed to the op once */
	int warn;
} sum_res;

/* Rvector_summarization.c:97-165 */
static int init_res(const sum_op *op, sum_res *r)
{
	memset(r, 0, sizeof(*r));
	r->status = ST_SET;
	switch (op->opcode) {
	    case ORC_OP_ANYNA: case ORC_OP_ANY:
		r->out_Rtype = ORC_LGL; r->buf.i[0] = 0; return 0;
	    case ORC_OP_COUNTNAS:
		r->out_Rtype = ORC_DBL; r->buf.d[0] = 0.0; return 0;
	    case ORC_OP_ALL:
		r->out_Rtype = ORC_LGL; r->buf.i[0] = 1; r->one_zero = 1;
		return 0;
	    case ORC_OP_SUM: case ORC_OP_MEAN:
	    case ORC_OP_CENTERED_X2_SUM: case ORC_OP_VAR1: case ORC_OP_SD1:
		r->out_Rtype = ORC_DBL; r->buf.d[0] = 0.0; return 0;
	    case ORC_OP_PROD:
		r->out_Rtype = ORC_DBL; r->buf.d[0] = 1.0; r->one_zero = 1;
		return 0;
	    case ORC_OP_SUM_X_X2: case ORC_OP_VAR2: case ORC_OP_SD2:
		r->out_Rtype = ORC_DBL; r->buf.d[0] = r->buf.d[1] = 0.0;
		return 0;
	    case ORC_OP_MIN: case ORC_OP_MAX: case ORC_OP_RANGE:
		break;
	    default:
		return fail("unknown opcode %d", op->opcode);
	}
	r->one_zero = 1;
	if (op->in_Rtype == ORC_INT || op->in_Rtype == ORC_LGL) {
		r->out_Rtype = ORC_INT;
		r->status = ST_NOT_SET;
		return 0;
	}
	r->out_Rtype = ORC_DBL;
	r->buf.d[0] = op->opcode == ORC_OP_MAX ? -INFINITY : INFINITY;
	r->buf.d[1] = -INFINITY;
	return 0;
}

/* int kernels: Rvector_summarization.c:177-186, 230-239, 286-342, 345-363,
   400-418, 455-482, 518-537, 570-587, 622-642, 677-700 */
static int feed_ints(const int *x, int n, const sum_op *op, sum_res *r)
{
	int narm = op->na_rm;
	switch (op->opcode) {
	    case ORC_OP_ANYNA:
		for (int i = 0; i < n; i++)
			if (x[i] == NA_INT) { r->buf.i[0] = 1; return ST_BREAK; }
		return ST_SET;
	    case ORC_OP_COUNTNAS: {
		double c = r->buf.d[0];
		for (int i = 0; i < n; i++)
			if (x[i] == NA_INT) c++;
		r->buf.d[0] = c;
		return ST_SET;
	    }
	    case ORC_OP_ANY: case ORC_OP_ALL: {
		int saw_NA = 0, hit = op->opcode == ORC_OP_ANY;
		for (int i = 0; i < n; i++) {
			if (x[i] == NA_INT) {
				if (narm) r->in_nacount++; else saw_NA = 1;
				continue;
			}
			if ((x[i] != 0) == hit) {
				r->buf.i[0] = hit;
				return ST_BREAK;
			}
		}
		if (saw_NA) r->buf.i[0] = NA_INT;
		return ST_SET;
	    }
	    case ORC_OP_MIN: case ORC_OP_MAX: case ORC_OP_RANGE: {
		int st = r->status, lo = r->buf.i[0];
		int hi = op->opcode == ORC_OP_RANGE ? r->buf.i[1] : lo;
		for (int i = 0; i < n; i++) {
			int v = x[i];
			if (v == NA_INT) {
				if (narm) { r->in_nacount++; continue; }
				r->buf.i[0] = r->buf.i[1] = NA_INT;
				return ST_BREAK;
			}
			if (st == ST_NOT_SET) {
				lo = hi = v;
				st = ST_SET;
				continue;
			}
			if (v < lo) lo = v;
			if (v > hi) hi = v;
		}
		if (op->opcode == ORC_OP_MAX) {
			r->buf.i[0] = hi;
		} else {
			r->buf.i[0] = lo;
			r->buf.i[1] = hi;
		}
		return st;
	    }
	    case ORC_OP_SUM: case ORC_OP_MEAN: case ORC_OP_PROD:
	    case ORC_OP_CENTERED_X2_SUM: case ORC_OP_VAR1: case ORC_OP_SD1:
	    case ORC_OP_SUM_X_X2: case ORC_OP_VAR2: case ORC_OP_SD2: {
		double a0 = r->buf.d[0], a1 = r->buf.d[1];
		for (int i = 0; i < n; i++) {
			if (x[i] == NA_INT) {
				if (narm) { r->in_nacount++; continue; }
				r->buf.d[0] = NA_REAL;
				if (op->opcode == ORC_OP_SUM_X_X2 ||
				    op->opcode == ORC_OP_VAR2 ||
				    op->opcode == ORC_OP_SD2)
					r->buf.d[1] = NA_REAL;
				return ST_BREAK;
			}
			double v = (double) x[i];
			switch (op->opcode) {
			    case ORC_OP_SUM: case ORC_OP_MEAN:
				a0 += v; break;
			    case ORC_OP_PROD:
				a0 *= v; break;
			    case ORC_OP_SUM_X_X2: case ORC_OP_VAR2:
			    case ORC_OP_SD2:
				a0 += v; a1 += v * v; break;
			    default: {
				double d = v - op->center;
				a0 += d * d;
			    }
			}
		}
		r->buf.d[0] = a0;
		if (op->opcode == ORC_OP_SUM_X_X2 ||
		    op->opcode == ORC_OP_VAR2 || op->opcode == ORC_OP_SD2)
			r->buf.d[1] = a1;
		return ST_SET;
	    }
	}
	return ST_SET;
}

/* double kernels: Rvector_summarization.c:188-197, 241-250, 365-397,
   420-452, 485-516, 540-567, 590-619, 645-674, 703-734.
   Shared rule: NA stops everything; NaN sticks (later finite values are
   ignored) but a later NA still wins; na_rm counts and skips both. */
static int feed_doubles(const double *x, int n, const sum_op *op, sum_res *r)
{
	int narm = op->na_rm, oc = op->opcode;
	if (oc == ORC_OP_ANYNA) {
		for (int i = 0; i < n; i++)
			if (isnan(x[i])) { r->buf.i[0] = 1; return ST_BREAK; }
		return ST_SET;
	}
	if (oc == ORC_OP_COUNTNAS) {
		double c = r->buf.d[0];
		for (int i = 0; i < n; i++)
			if (isnan(x[i])) c++;
		r->buf.d[0] = c;
		return ST_SET;
	}
	int two = oc == ORC_OP_RANGE || oc == ORC_OP_SUM_X_X2 ||
		  oc == ORC_OP_VAR2 || oc == ORC_OP_SD2;
	double a0 = r->buf.d[0], a1 = r->buf.d[1];
	int live = !is_R_NaN(a0);
	for (int i = 0; i < n; i++) {
		double v = x[i];
		if (isnan(v)) {
			if (narm) { r->in_nacount++; continue; }
			if (is_R_NA(v)) {
				r->buf.d[0] = NA_REAL;
				if (two) r->buf.d[1] = NA_REAL;
				return ST_BREAK;
			}
			a0 = v;
			if (two) a1 = v;
			live = 0;
			continue;
		}
		if (!live)
			continue;
		switch (oc) {
		    case ORC_OP_MIN: if (v < a0) a0 = v; break;
		    case ORC_OP_MAX: if (v > a0) a0 = v; break;
		    case ORC_OP_RANGE:
			if (v < a0) a0 = v;
			if (v > a1) a1 = v;
			break;
		    case ORC_OP_SUM: case ORC_OP_MEAN: a0 += v; break;
		    case ORC_OP_PROD: a0 *= v; break;
		    case ORC_OP_SUM_X_X2: case ORC_OP_VAR2: case ORC_OP_SD2:
			a0 += v; a1 += v * v; break;
		    default: {   /* centered_X2_sum, var1, sd1 */
			double d = v - op->center;
			a0 += d * d;
		    }
		}
	}
	r->buf.d[0] = a0;
	if (two) r->buf.d[1] = a1;
	return ST_SET;
}

/* A run of n implicit ones (lacunar leaf): Rvector_summarization.c:742-825,
   including the "+= 1.0" of the SUM_X_X2 family (:817-820). */
static int feed_ones(int n, const sum_op *op, sum_res *r)
{
	if (n == 0)
		return r->status;
	int is_int = op->in_Rtype != ORC_DBL;
	switch (op->opcode) {
	    case ORC_OP_ANYNA: case ORC_OP_COUNTNAS: case ORC_OP_ALL:
	    case ORC_OP_PROD:
		return ST_SET;
	    case ORC_OP_ANY:
		r->buf.i[0] = 1;
		return ST_BREAK;
	    case ORC_OP_MIN: case ORC_OP_MAX: case ORC_OP_RANGE: {
		int want_lo = op->opcode != ORC_OP_MAX;
		int want_hi = op->opcode != ORC_OP_MIN;
		int hi_slot = op->opcode == ORC_OP_RANGE ? 1 : 0;
		if (is_int) {
			if (r->status == ST_NOT_SET) {
				r->buf.i[0] = r->buf.i[hi_slot] = 1;
			} else {
				if (want_lo && r->buf.i[0] > 1)
					r->buf.i[0] = 1;
				if (want_hi && r->buf.i[hi_slot] < 1)
					r->buf.i[hi_slot] = 1;
			}
		} else {
			if (want_lo && r->buf.d[0] > 1.0)
				r->buf.d[0] = 1.0;
			if (want_hi && r->buf.d[hi_slot] < 1.0)
				r->buf.d[hi_slot] = 1.0;
		}
		return ST_SET;
	    }
	    case ORC_OP_SUM: case ORC_OP_MEAN:
		r->buf.d[0] += (double) n;
		return ST_SET;
	    case ORC_OP_CENTERED_X2_SUM: case ORC_OP_VAR1: case ORC_OP_SD1: {
		double d = 1.0 - op->center;
		r->buf.d[0] += d * d * n;
		return ST_SET;
	    }
	    default:
		r->buf.d[0] += 1.0;
		r->buf.d[1] += 1.0;
		return ST_SET;
	}
}

static void feed_values(const void *x, int n, const sum_op *op, sum_res *r)
{
	int st = op->in_Rtype == ORC_DBL ?
		feed_doubles((const double *) x, n, op, r) :
		feed_ints((const int *) x, n, op, r);
	r->status = st;
	if (st == ST_BREAK)
		r->one_zero = 0;
}

/* SparseArray_summarization.c:15-30 */
static void feed_leaf(const leaf_t *lf, const sum_op *op, sum_res *r)
{
	r->in_length += lf->len;
	r->in_nzcount += lf->n;
	if (lf->n == 0)
		return;
	if (lf->val == NULL) {
		int st = feed_ones(lf->n, op, r);
		r->status = st;
		if (st == ST_BREAK)
			r->one_zero = 0;
	} else {
		feed_values(lf->val, lf->n, op, r);
	}
}

/* Rvector_summarization.c:1078-1177 (zero background and, for NaArray objects,
   NA background: the implicit values count as NAs, :1086-1106) */
static int finish_res(sum_res *r, const sum_op *op)
{
	if (r->status == ST_BREAK)
		return 0;
	int oc = op->opcode;
	int64_t zerocount = r->in_length - r->in_nzcount;
	if (oc == ORC_OP_COUNTNAS) {
		if (op->na_bg)
			r->buf.d[0] += (double) zerocount;
		return 0;
	}
	int64_t n_eff = r->in_length;
	if (op->na_rm) {
		if (op->na_bg)
			n_eff = r->in_nzcount;
		n_eff -= r->in_nacount;
	}
	if (zerocount != 0 && op->na_bg) {
		/* summarize_one_NA(), :1033-1076: nothing to do under na.rm */
		if (!op->na_rm) {
			static const int iNA = NA_INT;
			const double dNA = NA_REAL;
			sum_op op0 = *op;
			op0.na_rm = 0;
			r->status = op->in_Rtype == ORC_DBL ?
				feed_doubles(&dNA, 1, &op0, r) : feed_ints(&iNA, 1, &op0, r);
			if (r->status == ST_BREAK)
				return 0;
		}
	} else if (zerocount != 0 && r->one_zero) {
		static const int i0 = 0;
		static const double d0 = 0.0;
		int64_t keep = r->in_nacount;
		r->status = op->in_Rtype == ORC_DBL ?
			feed_doubles(&d0, 1, op, r) : feed_ints(&i0, 1, op, r);
		r->in_nacount = keep;
	}
	if (r->status == ST_NOT_SET) {
		/* int min/max/range of nothing: NA + warning, :1108-1128 */
		r->buf.i[0] = r->buf.i[1] = NA_INT;
		r->warn = 1;
		r->status = ST_SET;
		return 0;
	}
	switch (oc) {
	    case ORC_OP_MEAN:
		r->buf.d[0] /= (double) n_eff;
		break;
	    case ORC_OP_CENTERED_X2_SUM: case ORC_OP_VAR1: case ORC_OP_SD1:
		if (!op->na_bg)
			r->buf.d[0] += op->center * op->center * zerocount;
		if (oc == ORC_OP_CENTERED_X2_SUM)
			break;
		if (n_eff <= 1) {
			r->buf.d[0] = NA_REAL;
			break;
		}
		r->buf.d[0] /= (n_eff - 1.0);
		if (oc == ORC_OP_SD1)
			r->buf.d[0] = sqrt(r->buf.d[0]);
		break;
	    case ORC_OP_VAR2: case ORC_OP_SD2: {
		if (n_eff <= 1) {
			r->buf.d[0] = NA_REAL;
			break;
		}
		double s = r->buf.d[0], s2 = r->buf.d[1];
		double v = (s2 - s * s / n_eff) / (n_eff - 1.0);
		r->buf.d[0] = oc == ORC_OP_SD2 ? sqrt(v) : v;
		break;
	    }
	}
	return 0;
}

/* One pass over leaves [first, first+count), stopping at a breaking value:
   SparseArray_summarization.c:38-68 */
static void feed_leaf_range(const orc_svt *x, int64_t first, int64_t count,
			    const sum_op *op, sum_res *r)
{
	for (int64_t j = first; j < first + count; j++) {
		leaf_t lf = get_leaf(x, j);
		feed_leaf(&lf, op, r);
		if (r->status == ST_BREAK)
			return;
	}
}

/* SparseArray_summarization.c:70-109: var1/sd1/centered_X2_sum without a
   center take a full "mean" pass first. */
static int summarize_leaf_range(const orc_svt *x, int64_t first,
				int64_t count, const sum_op *op_in,
				sum_res *r)
{
	sum_op op = *op_in;
	if ((op.opcode == ORC_OP_CENTERED_X2_SUM ||
	     op.opcode == ORC_OP_VAR1 || op.opcode == ORC_OP_SD1) &&
	    isnan(op.center))
	{
		sum_op mop = op;
		sum_res mr;
		mop.opcode = ORC_OP_MEAN;
		if (init_res(&mop, &mr))
			return -1;
		feed_leaf_range(x, first, count, &mop, &mr);
		finish_res(&mr, &mop);
		op.center = mr.buf.d[0];
	}
	if (init_res(&op, r))
		return -1;
	feed_leaf_range(x, first, count, &op, r);
	return finish_res(r, &op);
}

static int check_op_type(int opcode, int Rtype)
{
	/* Rvector_summarization.c:19-78 */
	if (Rtype != ORC_LGL && Rtype != ORC_INT && Rtype != ORC_DBL)
		return fail("does not support SparseArray objects "
			    "of this type()");
	if ((opcode == ORC_OP_ANY || opcode == ORC_OP_ALL) &&
	    Rtype == ORC_DBL)
		return fail("any()/all() does not support SparseArray "
			    "objects of type() \"double\"");
	if (opcode < ORC_OP_ANYNA || opcode > ORC_OP_SD2)
		return fail("'op' must be one of: \"anyNA\", \"countNAs\", ...");
	return 0;
}

/* SparseArray_summarization.c:112-142 */
int orc_summarize_SVT(const orc_svt *x, int opcode, int na_rm, double center,
		      double *out_d, int *out_i, int *out_Rtype, int *warn)
{
	if (check_op_type(opcode, x->Rtype))
		return -1;
	sum_op op = { opcode, x->Rtype, na_rm, center, x->na_background };
	sum_res r;
	if (summarize_leaf_range(x, 0, x->nleaves, &op, &r))
		return -1;
	/* a 1-d SVT with ndim==1 has nleaves==1; ndim>=1 always */
	*out_Rtype = r.out_Rtype;
	out_d[0] = r.buf.d[0]; out_d[1] = r.buf.d[1];
	out_i[0] = r.buf.i[0]; out_i[1] = r.buf.i[1];
	*warn = r.warn;
	return 0;
}

/* ========================================================================
 * colStats -- SparseArray_matrixStats.c:26-31, 179-284
 */
int orc_colStats_out_Rtype(int opcode, int in_Rtype)
{
	sum_op op = { opcode, in_Rtype, 0, 0.0, 0 };
	sum_res r;
	if (init_res(&op, &r))
		return -1;
	return r.out_Rtype;
}

int orc_colStats_SVT(const orc_svt *x, int opcode, int na_rm, double center,
		     int dims, void *out, int *warn)
{
	if (check_op_type(opcode, x->Rtype))
		return -1;
	if (dims < 1 || dims > x->ndim)
		return fail("'dims' must be >= 1 and <= %d", x->ndim);
	sum_op op = { opcode, x->Rtype, na_rm, center, x->na_background };
	int out_Rtype = orc_colStats_out_Rtype(opcode, x->Rtype);
	/* one result per generalized column = 'inner' consecutive leaves */
	int64_t inner = 1, nout = 1;
	for (int a = 1; a < dims; a++)
		inner *= x->dim[a];
	for (int a = dims; a < x->ndim; a++)
		nout *= x->dim[a];
	int any_warn = 0, any_err = 0;
	/* when dims == ndim there is a single result over all leaves; with
	   ndim == 1 and dims == 1, nleaves == 1 */
	#pragma omp parallel for schedule(static)
	for (int64_t g = 0; g < nout; g++) {
		sum_res r;
		if (summarize_leaf_range(x, g * inner, inner, &op, &r)) {
			any_err = 1;
			continue;
		}
		if (r.warn)
			any_warn = 1;
		if (out_Rtype == ORC_DBL)
			((double *) out)[g] = r.buf.d[0];
		else
			((int *) out)[g] = r.buf.i[0];
	}
	*warn = any_warn;
	return any_err ? -1 : 0;
}

/* ========================================================================
 * rowStats -- SparseArray_matrixStats.c:303-1205
 */

/* :303-324, :326-350 */
static void upd_int_minmax(int v, int narm, int *out, int not_set, int is_min)
{
	if (narm) {
		if (v == NA_INT)
			return;
		if (*out == NA_INT) { *out = v; return; }
	} else {
		if (not_set || v == NA_INT) { *out = v; return; }
		if (*out == NA_INT)
			return;
	}
	if (is_min ? v < *out : v > *out)
		*out = v;
}

/* :352-380, :382-410 */
static void upd_double_minmax(double v, int narm, double *out, int not_set,
			      int is_min)
{
	if (narm) {
		if (isnan(v))
			return;
		if (is_R_NA(*out)) { *out = v; return; }
	} else {
		if (not_set || is_R_NA(v)) { *out = v; return; }
		if (isnan(*out))
			return;
		if (is_R_NaN(v)) { *out = v; return; }
	}
	if (is_min ? v < *out : v > *out)
		*out = v;
}

static inline int leaf_val_is_na(const leaf_t *lf, int Rtype, int k)
{
	if (Rtype == ORC_DBL)
		return isnan(((const double *) lf->val)[k]);
	return ((const int *) lf->val)[k] == NA_INT;
}

/* One leaf scattered into its slice of 'out': :498-772 (zero background) */
static void scatter_leaf(const leaf_t *lf, int Rtype, int opcode, int narm,
			 const double *center, void *out, int64_t *nzcvg)
{
	int n = lf->n;
	const int *off = lf->off;
	switch (opcode) {
	    case ORC_OP_ANYNA:       /* :498-514 */
		if (lf->val == NULL) return;
		for (int k = 0; k < n; k++)
			if (leaf_val_is_na(lf, Rtype, k))
				((int *) out)[off[k]] = 1;
		return;
	    case ORC_OP_COUNTNAS:    /* :516-535 */
		if (lf->val == NULL) return;
		for (int k = 0; k < n; k++)
			if (leaf_val_is_na(lf, Rtype, k))
				((double *) out)[off[k]]++;
		return;
	    case ORC_OP_MIN: case ORC_OP_MAX: {   /* :537-597 */
		int is_min = opcode == ORC_OP_MIN;
		for (int k = 0; k < n; k++) {
			int not_set = nzcvg[off[k]]++ == 0;
			if (Rtype == ORC_DBL) {
				double v = lf->val ?
					((const double *) lf->val)[k] : 1.0;
				upd_double_minmax(v, narm,
					(double *) out + off[k], not_set,
					is_min);
			} else {
				int v = lf->val ?
					((const int *) lf->val)[k] : 1;
				upd_int_minmax(v, narm,
					(int *) out + off[k], not_set, is_min);
			}
		}
		return;
	    }
	    case ORC_OP_SUM: {       /* :599-634 with :412-433 */
		double *o = (double *) out;
		if (lf->val == NULL) {
			for (int k = 0; k < n; k++)
				o[off[k]] += 1.0;
			return;
		}
		for (int k = 0; k < n; k++) {
			double v;
			if (Rtype == ORC_DBL) {
				v = ((const double *) lf->val)[k];
				if (narm && isnan(v)) continue;
			} else {
				int iv = ((const int *) lf->val)[k];
				if (iv == NA_INT) {
					if (narm) continue;
					v = NA_REAL;
				} else {
					v = (double) iv;
				}
			}
			o[off[k]] += v;
		}
		return;
	    }
	    case ORC_OP_CENTERED_X2_SUM: {   /* :636-696 */
		double *o = (double *) out;
		for (int k = 0; k < n; k++) {
			int i = off[k];
			double c = center ? center[i] : 0.0, v;
			if (lf->val == NULL) {
				/* lacunar: adds 1 - 2c, :648-654 */
				double t = 1.0;
				if (center) t -= 2 * center[i];
				o[i] += t;
				continue;
			}
			if (Rtype == ORC_DBL) {
				v = ((const double *) lf->val)[k];
				if (narm && isnan(v)) { o[i] -= c * c; continue; }
			} else {
				int iv = ((const int *) lf->val)[k];
				if (iv == NA_INT) {
					if (narm) { o[i] -= c * c; continue; }
					v = NA_REAL;
				} else {
					v = (double) iv;
				}
			}
			o[i] += v * (v - 2 * c);
		}
		return;
	    }
	}
}

int orc_rowStats_SVT(const orc_svt *x, int opcode, int na_rm,
		     const double *center, int dims, void *out, int *warn)
{
	*warn = 0;
	if (check_op_type(opcode, x->Rtype))
		return -1;
	const int na_bg = x->na_background != 0;
	if (na_bg && opcode == ORC_OP_CENTERED_X2_SUM)   /* :639-642 */
		return fail("operation not yet supported on NaArray objects");
	if (na_bg && opcode == ORC_OP_ANYNA) {
		/* SVT_rowAnyNAs(), :880-910: count the NAs, then compare with 0 */
		int64_t inner_ = 1;
		for (int a = 1; a < dims && a < x->ndim; a++)
			inner_ *= x->dim[a];
		int64_t n_ = inner_ * x->dim[0];
		double *cnt = (double *) malloc(sizeof(double) * (n_ > 0 ? n_ : 1));
		int rc = orc_rowStats_SVT(x, ORC_OP_COUNTNAS, na_rm, center, dims, cnt, warn);
		if (rc == 0)
			for (int64_t i = 0; i < n_; i++)
				((int *) out)[i] = cnt[i] != 0.0;
		free(cnt);
		return rc;
	}
	if (dims < 1 || dims > x->ndim - 1)
		return fail("'dims' must be >= 1 and <= %d", x->ndim - 1);
	if (opcode != ORC_OP_COUNTNAS && opcode != ORC_OP_ANYNA &&
	    opcode != ORC_OP_MIN && opcode != ORC_OP_MAX &&
	    opcode != ORC_OP_SUM && opcode != ORC_OP_CENTERED_X2_SUM)
		return fail("operation not supported");
	int out_Rtype = orc_colStats_out_Rtype(opcode, x->Rtype);
	int dim0 = x->dim[0];
	int64_t inner = 1, nstrata = 1;   /* :1100-1118 */
	for (int a = 1; a < dims; a++)
		inner *= x->dim[a];
	for (int a = dims; a < x->ndim; a++)
		nstrata *= x->dim[a];
	int64_t out_len = inner * dim0;
	if (out_len == 0)
		return 0;
	int64_t *nzcvg = NULL;

	/* initialisation: :840-1060 */
	switch (opcode) {
	    case ORC_OP_COUNTNAS:   /* :856-878: an NaArray starts from nstrata */
		for (int64_t i = 0; i < out_len; i++)
			((double *) out)[i] = na_bg ? (double) nstrata : 0.0;
		break;
	    case ORC_OP_SUM:
		memset(out, 0, sizeof(double) * out_len);
		break;
	    case ORC_OP_ANYNA:
		memset(out, 0, sizeof(int) * out_len);
		break;
	    case ORC_OP_CENTERED_X2_SUM:
		for (int64_t i = 0; i < out_len; i++) {
			double c = center ? center[i] : 0.0;
			((double *) out)[i] = center ? c * c * nstrata : 0.0;
		}
		break;
	    default:   /* min / max, :963-1019 */
		if (nstrata == 0) {
			for (int64_t i = 0; i < out_len; i++) {
				if (out_Rtype == ORC_DBL)
					((double *) out)[i] =
						opcode == ORC_OP_MIN ?
						INFINITY : -INFINITY;
				else
					((int *) out)[i] = NA_INT;
			}
			if (out_Rtype != ORC_DBL)
				*warn = 1;
			return 0;
		}
		for (int64_t i = 0; i < out_len; i++) {
			/* uninitialised in the reference unless na_rm */
			if (out_Rtype == ORC_DBL)
				((double *) out)[i] = na_rm ? NA_REAL : 0.0;
			else
				((int *) out)[i] = na_rm ? NA_INT : 0;
		}
		nzcvg = (int64_t *) calloc(out_len, sizeof(int64_t));
	}
	/* NA background, na.rm=FALSE, anything but countNAs: a NULL leaf / subtree turns
	   its whole slice of 'out' into NAs (update_out_for_rowStats_NULL, :756-768) */
	const int null_gives_na = na_bg && !na_rm && opcode != ORC_OP_COUNTNAS;
	if (nstrata != 0 && (!x->svt_is_null || null_gives_na)) {
		/* the DFS of :774-829 visits leaves in flat order */
		for (int64_t j = 0; j < x->nleaves; j++) {
			leaf_t lf = get_leaf(x, j);
			int64_t base = (j % inner) * dim0;
			size_t esz = out_Rtype == ORC_DBL ? 8 : 4;
			if (lf.n == 0) {
				if (null_gives_na)
					for (int i = 0; i < dim0; i++) {
						if (out_Rtype == ORC_DBL)
							((double *) out)[base + i] = NA_REAL;
						else
							((int *) out)[base + i] = NA_INT;
					}
				continue;
			}
			if (na_bg && opcode == ORC_OP_COUNTNAS) {
				/* :516-535: every stored non-NA value takes one off */
				for (int k = 0; k < lf.n; k++)
					if (lf.val == NULL || !leaf_val_is_na(&lf, x->Rtype, k))
						((double *) out)[base + lf.off[k]]--;
				continue;
			}
			if (na_bg && opcode == ORC_OP_SUM && !na_rm) {
				/* rowSums(<NaArray>, na.rm=FALSE), :612-634: walk all the
				   positions, the implicit ones are NAs */
				double *o = (double *) out + base;
				int k = 0;
				for (int i = 0; i < dim0; i++) {
					if (k < lf.n && lf.off[k] == i) {
						double v;
						if (x->Rtype == ORC_DBL) {
							v = ((const double *) lf.val)[k];
						} else {
							int iv = ((const int *) lf.val)[k];
							v = iv == NA_INT ? NA_REAL : (double) iv;
						}
						o[i] += v;
						k++;
					} else {
						o[i] = NA_REAL;
					}
				}
				continue;
			}
			scatter_leaf(&lf, x->Rtype, opcode, na_rm,
				     center ? center + base : NULL,
				     (char *) out + base * esz,
				     nzcvg ? nzcvg + base : NULL);
		}
	}
	if (opcode == ORC_OP_MIN || opcode == ORC_OP_MAX) {
		/* :914-961: cells not covered nstrata times also see a 0 */
		int is_min = opcode == ORC_OP_MIN;
		for (int64_t i = 0; i < out_len; i++) {
			int64_t cv = nzcvg[i];
			if (out_Rtype == ORC_DBL) {
				double *o = (double *) out + i;
				if (cv < nstrata)
					upd_double_minmax(na_bg ? NA_REAL : 0.0, na_rm, o,
							  cv == 0, is_min);
				if (na_rm && is_R_NA(*o))
					*o = is_min ? INFINITY : -INFINITY;
			} else {
				int *o = (int *) out + i;
				if (cv < nstrata)
					upd_int_minmax(na_bg ? NA_INT : 0, na_rm, o,
						       cv == 0, is_min);
				if (na_rm && *o == NA_INT)
					*warn = 1;
			}
		}
		free(nzcvg);
	}
	return 0;
}

/* ========================================================================
 * rowsum / colsum -- rowsum_methods.c
 */

/* rowsum_methods.c:15-37 */
static int check_group(const int *group, int n, int ngroup)
{
	for (int i = 0; i < n; i++) {
		int g = group[i];
		if (g == NA_INT) {
			if (ngroup < 1)
				return fail("'ngroup' must be >= 1 when "
					    "'group' contains missing values");
		} else if (g < 1 || g > ngroup) {
			return fail("all non-NA values in 'group' must "
				    "be >= 1 and <= 'ngroup'");
		}
	}
	return 0;
}

/* S4Vectors safe_int_add() (third-party; restated from its published
   behaviour: NA in -> NA out; result outside [-INT_MAX, INT_MAX] sets the
   overflow flag and gives NA).  Overflow parity is UNPINNED. */
static int safe_int_add(int x, int y, int *ovflow)
{
	if (x == NA_INT || y == NA_INT)
		return NA_INT;
	if ((y > 0 && x > INT_MAX - y) || (y < 0 && x < -INT_MAX - y)) {
		*ovflow = 1;
		return NA_INT;
	}
	return x + y;
}

static int too_many_groups(int64_t a, int64_t b)
{
	if (a * b > INT_MAX)
		return fail("too many groups (matrix of sums will be "
			    "too big)");
	return 0;
}

/* rowsum_methods.c:44-84 */
static void rowsum_one_col(const int *off, const void *val, int n, int Rtype,
			   const int *group, int narm, void *out, int ngroup,
			   int *ovflow)
{
	for (int k = 0; k < n; k++) {
		int g = group[off[k]];
		if (g == NA_INT)
			g = ngroup;
		g--;
		if (Rtype == ORC_DBL) {
			double v = 1.0;
			if (val) {
				v = ((const double *) val)[k];
				if (narm && isnan(v))
					continue;
			}
			((double *) out)[g] += v;
		} else {
			int v = 1;
			if (val) {
				v = ((const int *) val)[k];
				if (narm && v == NA_INT)
					continue;
			}
			int *o = (int *) out + g;
			*o = safe_int_add(*o, v, ovflow);
		}
	}
}

/* rowsum_methods.c:281-325 with :86-125 */
int orc_rowsum_SVT(const orc_svt *x, const int *group, int ngroup, int na_rm,
		   void *out, int *ovflow)
{
	*ovflow = 0;
	if (x->ndim != 2)
		return fail("input object must have 2 dimensions");
	if (x->Rtype != ORC_DBL && x->Rtype != ORC_INT)
		return fail("rowsum() and colsum() do not support "
			    "SVT_SparseMatrix objects of this type");
	int nrow = x->dim[0], ncol = x->dim[1];
	if (check_group(group, nrow, ngroup) ||
	    too_many_groups(ngroup, ncol))
		return -1;
	size_t esz = x->Rtype == ORC_DBL ? 8 : 4;
	memset(out, 0, esz * (size_t) ngroup * ncol);
	if (x->svt_is_null)
		return 0;
	for (int j = 0; j < ncol; j++) {
		leaf_t lf = get_leaf(x, j);
		if (lf.n == 0)
			continue;
		rowsum_one_col(lf.off, lf.val, lf.n, x->Rtype, group, na_rm,
			       (char *) out + esz * (size_t) j * ngroup,
			       ngroup, ovflow);
	}
	return 0;
}

/* rowsum_methods.c:146-199 */
static void add_leaf_to_col(const int *off, const void *val, int n, int Rtype,
			    int narm, void *out, int *ovflow)
{
	for (int k = 0; k < n; k++) {
		if (Rtype == ORC_DBL) {
			double v = 1.0;
			if (val) {
				v = ((const double *) val)[k];
				if (narm && isnan(v))
					continue;
			}
			((double *) out)[off[k]] += v;
			continue;
		}
		int *o = (int *) out + off[k];
		if (*o == NA_INT)
			continue;
		int v = 1;
		if (val) {
			v = ((const int *) val)[k];
			if (v == NA_INT) {
				if (!narm)
					*o = NA_INT;
				continue;
			}
		}
		double y = (double) *o + v;
		if (-INT_MAX <= y && y <= INT_MAX) {
			*o = (int) y;
		} else {
			*ovflow = 1;
			*o = NA_INT;
		}
	}
}

/* rowsum_methods.c:363-401 with :204-255 */
int orc_colsum_SVT(const orc_svt *x, const int *group, int ngroup, int na_rm,
		   void *out, int *ovflow)
{
	*ovflow = 0;
	if (x->ndim != 2)
		return fail("input object must have 2 dimensions");
	if (x->Rtype != ORC_DBL && x->Rtype != ORC_INT)
		return fail("rowsum() and colsum() do not support "
			    "SVT_SparseMatrix objects of this type");
	int nrow = x->dim[0], ncol = x->dim[1];
	if (check_group(group, ncol, ngroup) ||
	    too_many_groups(nrow, ngroup))
		return -1;
	size_t esz = x->Rtype == ORC_DBL ? 8 : 4;
	memset(out, 0, esz * (size_t) nrow * ngroup);
	if (x->svt_is_null)
		return 0;
	for (int j = 0; j < ncol; j++) {
		leaf_t lf = get_leaf(x, j);
		if (lf.n == 0)
			continue;
		int g = group[j];
		if (g == NA_INT)
			g = ngroup;
		g--;
		add_leaf_to_col(lf.off, lf.val, lf.n, x->Rtype, na_rm,
				(char *) out + esz * (size_t) g * nrow,
				ovflow);
	}
	return 0;
}

/* rowsum_methods.c:127-139, :328-356 */
int orc_rowsum_dgCMatrix(int nrow, int ncol, const double *xx, const int *xi,
			 const int *xp, const int *group, int ngroup,
			 int na_rm, double *out)
{
	int ov = 0;
	if (check_group(group, nrow, ngroup) ||
	    too_many_groups(ngroup, ncol))
		return -1;
	memset(out, 0, sizeof(double) * (size_t) ngroup * ncol);
	for (int j = 0; j < ncol; j++)
		rowsum_one_col(xi + xp[j], xx + xp[j], xp[j + 1] - xp[j],
			       ORC_DBL, group, na_rm,
			       out + (size_t) j * ngroup, ngroup, &ov);
	return 0;
}

/* rowsum_methods.c:257-273, :404-439 */
int orc_colsum_dgCMatrix(int nrow, int ncol, const double *xx, const int *xi,
			 const int *xp, const int *group, int ngroup,
			 int na_rm, double *out)
{
	int ov = 0;
	if (check_group(group, ncol, ngroup) ||
	    too_many_groups(nrow, ngroup))
		return -1;
	memset(out, 0, sizeof(double) * (size_t) nrow * ngroup);
	for (int j = 0; j < ncol; j++) {
		int g = group[j];
		if (g == NA_INT)
			g = ngroup;
		g--;
		add_leaf_to_col(xi + xp[j], xx + xp[j], xp[j + 1] - xp[j],
				ORC_DBL, na_rm, out + (size_t) g * nrow, &ov);
	}
	return 0;
}

/* ========================================================================
 * Column statistics of a dgCMatrix -- sparseMatrix_utils.c:8-223.  Only the
 * Dim, x and p slots are read (:108-113, :145-150, :207-212).
 */

/* sparseMatrix_utils.c:15-39 */
static double dgc_min_double(const double *x, int x_len, int narm, int start_on_zero)
{
	double min = start_on_zero ? 0.0 : INFINITY;
	int min_is_NaN = 0;
	for (int i = 0; i < x_len; i++) {
		double xi = x[i];
		if (is_R_NA(xi)) {
			if (narm)
				continue;
			return NA_REAL;
		}
		if (min_is_NaN)
			continue;
		if (is_R_NaN(xi)) {
			if (narm)
				continue;
			min = xi;
			min_is_NaN = 1;
			continue;
		}
		if (xi < min)
			min = xi;
	}
	return min;
}

/* sparseMatrix_utils.c:41-65 */
static double dgc_max_double(const double *x, int x_len, int narm, int start_on_zero)
{
	double max = start_on_zero ? 0.0 : -INFINITY;
	int max_is_NaN = 0;
	for (int i = 0; i < x_len; i++) {
		double xi = x[i];
		if (is_R_NA(xi)) {
			if (narm)
				continue;
			return NA_REAL;
		}
		if (max_is_NaN)
			continue;
		if (is_R_NaN(xi)) {
			if (narm)
				continue;
			max = xi;
			max_is_NaN = 1;
			continue;
		}
		if (xi > max)
			max = xi;
	}
	return max;
}

/* sparseMatrix_utils.c:67-103 */
static void dgc_minmax_double(const double *x, int x_len, int narm, int start_on_zero,
			      double *min, double *max)
{
	double tmp_min, tmp_max;
	if (start_on_zero) {
		tmp_min = tmp_max = 0.0;
	} else {
		tmp_min = INFINITY;
		tmp_max = -INFINITY;
	}
	int is_NaN = 0;
	for (int i = 0; i < x_len; i++) {
		double xi = x[i];
		if (is_R_NA(xi)) {
			if (narm)
				continue;
			*min = *max = NA_REAL;
			return;
		}
		if (is_NaN)
			continue;
		if (is_R_NaN(xi)) {
			if (narm)
				continue;
			tmp_min = tmp_max = xi;
			is_NaN = 1;
			continue;
		}
		if (xi < tmp_min)
			tmp_min = xi;
		if (xi > tmp_max)
			tmp_max = xi;
	}
	*min = tmp_min;
	*max = tmp_max;
}

/* C_colExtrema_dgCMatrix + C_colMins_dgCMatrix, sparseMatrix_utils.c:105-132 */
int orc_colMins_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
			  int na_rm, double *out)
{
	for (int j = 0; j < ncol; j++) {
		int offset = xp[j], nzcount = xp[j + 1] - offset;
		out[j] = dgc_min_double(xx + offset, nzcount, na_rm, nzcount < nrow);
	}
	return 0;
}

/* C_colMaxs_dgCMatrix, sparseMatrix_utils.c:134-138 */
int orc_colMaxs_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
			  int na_rm, double *out)
{
	for (int j = 0; j < ncol; j++) {
		int offset = xp[j], nzcount = xp[j + 1] - offset;
		out[j] = dgc_max_double(xx + offset, nzcount, na_rm, nzcount < nrow);
	}
	return 0;
}

/* C_colRanges_dgCMatrix, sparseMatrix_utils.c:143-166: ans is ncol x 2 */
int orc_colRanges_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
			    int na_rm, double *out)
{
	for (int j = 0; j < ncol; j++) {
		int offset = xp[j], nzcount = xp[j + 1] - offset;
		dgc_minmax_double(xx + offset, nzcount, na_rm, nzcount < nrow,
				  out + j, out + ncol + j);
	}
	return 0;
}

/* col_sum(), sparseMatrix_utils.c:173-188 */
static double dgc_col_sum(const double *x, int x_len, int nrow, int narm, int *sample_size)
{
	*sample_size = nrow;
	double sum = 0.0;
	for (int i = 0; i < x_len; i++) {
		double xi = x[i];
		if (narm && isnan(xi)) {
			(*sample_size)--;
			continue;
		}
		sum += xi;
	}
	return sum;
}

/* col_var(), sparseMatrix_utils.c:190-203 */
static double dgc_col_var(const double *x, int x_len, int nrow, int narm)
{
	int sample_size;
	double sum = dgc_col_sum(x, x_len, nrow, narm, &sample_size);
	double mean = sum / (double) sample_size;
	double sigma = mean * mean * (nrow - x_len);
	for (int i = 0; i < x_len; i++) {
		double xi = x[i];
		if (narm && isnan(xi))
			continue;
		double delta = xi - mean;
		sigma += delta * delta;
	}
	return sigma / (sample_size - 1.0);
}

/* C_colVars_dgCMatrix, sparseMatrix_utils.c:205-223 */
int orc_colVars_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
			  int na_rm, double *out)
{
	for (int j = 0; j < ncol; j++) {
		int offset = xp[j], nzcount = xp[j + 1] - offset;
		out[j] = dgc_col_var(xx + offset, nzcount, nrow, na_rm);
	}
	return 0;
}

/* ========================================================================
 * 2-D transposition -- C_transpose_2D_SVT, SparseArray_aperm.c:148-423.
 * The reference's three passes: count the nonzeros of every input row
 * (collect_stats_on_input_rows, :148-171), allocate the output leaves
 * (:366-385), scatter column by column (transpose_*_col, :177-241; visiting
 * the columns in ascending order is what leaves every output leaf sorted by
 * offset).  Output here: the CSC layout of t(x) (nrow + 1 pointers); lacunar
 * input leaves come out as explicit ones.
 */
int orc_transpose_2D_SVT(const orc_svt *x, int64_t *out_col_ptr,
			 int32_t *out_row_idx, void *out_val)
{
	if (x->ndim != 2)
		return fail("object to transpose must have exactly 2 dimensions");
	int nrow = x->dim[0], ncol = x->dim[1];
	size_t esz = x->Rtype == ORC_DBL ? 8 : 4;
	for (int i = 0; i <= nrow; i++)
		out_col_ptr[i] = 0;
	/* 1st pass: nzcount per input row */
	for (int j = 0; j < ncol; j++) {
		leaf_t lf = get_leaf(x, j);
		for (int k = 0; k < lf.n; k++)
			out_col_ptr[lf.off[k] + 1]++;
	}
	/* 2nd pass: "allocation" = where every output leaf starts */
	for (int i = 0; i < nrow; i++)
		out_col_ptr[i + 1] += out_col_ptr[i];
	/* 3rd pass: fill */
	int64_t *fill = (int64_t *) malloc(sizeof(int64_t) * (size_t) (nrow > 0 ? nrow : 1));
	if (fill == NULL)
		return fail("out of memory");
	memcpy(fill, out_col_ptr, sizeof(int64_t) * (size_t) nrow);
	for (int j = 0; j < ncol; j++) {
		leaf_t lf = get_leaf(x, j);
		for (int k = 0; k < lf.n; k++) {
			int64_t p = fill[lf.off[k]]++;
			out_row_idx[p] = j;
			if (esz == 8)
				((double *) out_val)[p] = lf.val ? ((const double *) lf.val)[k] : 1.0;
			else
				((int *) out_val)[p] = lf.val ? ((const int *) lf.val)[k] : 1;
		}
	}
	free(fill);
	return 0;
}

/* ========================================================================
 * aperm -- SparseArray_aperm.c.  The reference has a leaf-preserving fast
 * path (perm[1] == 1, :949-957) and a counting-sort path that shatters the
 * leaves (:892-929); both produce the unique SVT of base::aperm(dense): here
 * the nonzeros are keyed by their linear index in the permuted array and
 * sorted.
 */
typedef struct { unsigned long long key; int64_t leaf; int k; } aperm_ent;

static int aperm_cmp(const void *a, const void *b)
{
	unsigned long long x = ((const aperm_ent *) a)->key, y = ((const aperm_ent *) b)->key;
	return x < y ? -1 : x > y;
}

int orc_aperm_SVT(const orc_svt *x, const int *perm, int64_t *out_col_ptr,
		  int32_t *out_row_idx, void *out_val)
{
	int nd = x->ndim;
	if (nd < 1 || nd > 8)
		return fail("aperm: between 1 and 8 dimensions are supported");
	int seen[8] = {0};
	int64_t mul[8], m = 1, new_nl = 1;
	for (int a = 0; a < nd; a++) {
		if (perm[a] < 1 || perm[a] > nd || seen[perm[a] - 1])
			return fail("'perm' must be a permutation of 1:%d", nd);
		seen[perm[a] - 1] = 1;
		mul[perm[a] - 1] = m;
		m *= x->dim[perm[a] - 1];
		if (a >= 1)
			new_nl *= x->dim[perm[a] - 1];
	}
	int64_t new_dim0 = x->dim[perm[0] - 1];
	int64_t nnz = 0;
	if (!x->svt_is_null)
		for (int64_t j = 0; j < x->nleaves; j++)
			nnz += x->nzcount[j];
	aperm_ent *e = (aperm_ent *) malloc(sizeof(aperm_ent) * (nnz > 0 ? nnz : 1));
	int64_t n = 0;
	for (int64_t j = 0; j < x->nleaves && nnz > 0; j++) {
		leaf_t lf = get_leaf(x, j);
		unsigned long long base = 0;
		int64_t rest = j;
		for (int a = 1; a < nd; a++) {
			base += (unsigned long long) (rest % x->dim[a]) * (unsigned long long) mul[a];
			rest /= x->dim[a];
		}
		for (int k = 0; k < lf.n; k++) {
			e[n].key = base + (unsigned long long) lf.off[k] * (unsigned long long) mul[0];
			e[n].leaf = j;
			e[n].k = k;
			n++;
		}
	}
	qsort(e, (size_t) n, sizeof(aperm_ent), aperm_cmp);
	int64_t pos = 0;
	for (int64_t L = 0; L <= new_nl; L++) {
		unsigned long long lim = (unsigned long long) L * (unsigned long long) new_dim0;
		while (pos < n && e[pos].key < lim)
			pos++;
		out_col_ptr[L] = pos;
	}
	for (int64_t i = 0; i < n; i++) {
		leaf_t lf = get_leaf(x, e[i].leaf);
		out_row_idx[i] = (int32_t) (e[i].key % (unsigned long long) (new_dim0 > 0 ? new_dim0 : 1));
		if (x->Rtype == ORC_DBL)
			((double *) out_val)[i] = lf.val ? ((const double *) lf.val)[e[i].k] : 1.0;
		else
			((int *) out_val)[i] = lf.val ? ((const int *) lf.val)[e[i].k] : 1;
	}
	free(e);
	return 0;
}
