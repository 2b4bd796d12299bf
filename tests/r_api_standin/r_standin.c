/*
 * Test-only FUNCTIONAL stand-in for the part of R's C API that integration/svt_hip_glue.c (and the
 * reference helpers it calls) use: fake SEXPs on the C heap, attributes, symbols, a protection stack
 * that is counted, R_alloc() arenas, error() as a longjmp back to the harness, warning() into a log.
 * Written from R's documented API ("Writing R Extensions"); it contains no code of R or of the
 * reference.  It exists because the image has no R: tests/test_glue_executes.py links the glue against
 * it to RUN the 20 registered entry points on the CPU (the svt_* symbols bound to the oracle).  Nothing
 * here is shipped, linked into the product or used as an oracle.
 *
 * Harness entry points (called from Python through ctypes): sx_* below.
 */
#include "Rdefines.h"

#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SX_MAXATTR 12

struct SEXPREC {
	int type;
	R_xlen_t len;
	void *data;                     /* element storage; CHARSXP / SYMSXP: the C string */
	int nattr;
	SEXP attr_tag[SX_MAXATTR];
	SEXP attr_val[SX_MAXATTR];
};

#define CHARSXP 9

static struct SEXPREC nil_rec = { NILSXP, 0, NULL, 0, {0}, {0} };
SEXP R_NilValue = &nil_rec;
SEXP R_NaString, R_BlankString, R_DimSymbol, R_DimNamesSymbol, R_NamesSymbol, R_ClassSymbol;
double R_NaReal, R_NaN, R_PosInf, R_NegInf;
int R_NaInt = INT_MIN;

/* ---- bookkeeping ---------------------------------------------------------------------------- */
static int protect_depth, protect_max, protect_underflow;
static jmp_buf *err_jmp;
static char err_msg[2048];
static char warn_log[8192];
static int nwarn;
static void **arena;             /* R_alloc() blocks of the running call */
static size_t arena_n, arena_cap;
static SEXP *objs;               /* every SEXP ever made (freed by sx_reset) */
static size_t objs_n, objs_cap;
static SEXP *symtab;
static size_t sym_n, sym_cap;

static void *xmalloc(size_t n)
{
	void *p = calloc(n ? n : 1, 1);
	if (p == NULL) { fprintf(stderr, "r_standin: out of memory\n"); abort(); }
	return p;
}

static SEXP new_rec(int type, R_xlen_t len, size_t esz)
{
	SEXP s = (SEXP) xmalloc(sizeof(struct SEXPREC));
	s->type = type;
	s->len = len;
	s->data = xmalloc((size_t) (len > 0 ? len : 0) * esz + 16);
	if (objs_n == objs_cap) {
		objs_cap = objs_cap ? objs_cap * 2 : 1024;
		objs = (SEXP *) realloc(objs, objs_cap * sizeof(SEXP));
	}
	objs[objs_n++] = s;
	return s;
}

static size_t elt_size(SEXPTYPE t)
{
	switch (t) {
	case LGLSXP: case INTSXP: return 4;
	case REALSXP: return 8;
	case CPLXSXP: return 16;
	case RAWSXP: return 1;
	case STRSXP: case VECSXP: return sizeof(SEXP);
	}
	return 0;
}

static void init_once(void)
{
	static int done;
	if (done) return;
	done = 1;
	union { uint64_t u; double d; } na;
	na.u = 0x7FF00000000007A2ULL;                   /* R's NA_real_: a NaN whose low word is 1954 */
	R_NaReal = na.d;
	R_NaN = 0.0 / 0.0;
	R_PosInf = 1.0 / 0.0;
	R_NegInf = -1.0 / 0.0;
	R_NaString = Rf_mkChar("NA");
	R_BlankString = Rf_mkChar("");
	R_DimSymbol = Rf_install("dim");
	R_DimNamesSymbol = Rf_install("dimnames");
	R_NamesSymbol = Rf_install("names");
	R_ClassSymbol = Rf_install("class");
}

/* ---- the API -------------------------------------------------------------------------------- */
int R_IsNA(double x)
{
	union { double d; uint32_t w[2]; } u;
	u.d = x;
	return x != x && u.w[0] == 1954;
}
int R_IsNaN(double x) { return x != x && !R_IsNA(x); }
int R_finite(double x) { return x == x && x != R_PosInf && x != R_NegInf; }

int *INTEGER(SEXP s) { return (int *) s->data; }
int *LOGICAL(SEXP s) { return (int *) s->data; }
double *REAL(SEXP s) { return (double *) s->data; }
Rcomplex *COMPLEX(SEXP s) { return (Rcomplex *) s->data; }
Rbyte *RAW(SEXP s) { return (Rbyte *) s->data; }
void *DATAPTR(SEXP s) { return s->data; }
SEXP VECTOR_ELT(SEXP s, R_xlen_t i) { return ((SEXP *) s->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP s, R_xlen_t i, SEXP v) { ((SEXP *) s->data)[i] = v; return v; }
SEXP STRING_ELT(SEXP s, R_xlen_t i) { return ((SEXP *) s->data)[i]; }
void SET_STRING_ELT(SEXP s, R_xlen_t i, SEXP v) { ((SEXP *) s->data)[i] = v; }
int LENGTH(SEXP s) { return (int) s->len; }
R_xlen_t XLENGTH(SEXP s) { return s->len; }
int TYPEOF(SEXP s) { return s->type; }
SEXP ATTRIB(SEXP s) { return s->nattr ? s : R_NilValue; }
const char *CHAR(SEXP s) { return (const char *) s->data; }

SEXP Rf_allocVector(SEXPTYPE t, R_xlen_t n)
{
	init_once();
	SEXP s = new_rec((int) t, n, elt_size(t));
	if (t == VECSXP)
		for (R_xlen_t i = 0; i < n; i++) ((SEXP *) s->data)[i] = R_NilValue;
	else if (t == STRSXP)
		for (R_xlen_t i = 0; i < n; i++) ((SEXP *) s->data)[i] = R_BlankString;
	else if (n > 0)
		memset(s->data, 0xCD, (size_t) n * elt_size(t));   /* R does not initialise atomic vectors: a result the
								      glue forgets to fill must not look like zeros */
	return s;
}

SEXP Rf_setAttrib(SEXP s, SEXP tag, SEXP v)
{
	for (int i = 0; i < s->nattr; i++)
		if (s->attr_tag[i] == tag) {
			s->attr_val[i] = v;
			return v;
		}
	if (v == R_NilValue)
		return v;
	if (s->nattr == SX_MAXATTR) { fprintf(stderr, "r_standin: too many attributes\n"); abort(); }
	s->attr_tag[s->nattr] = tag;
	s->attr_val[s->nattr++] = v;
	return v;
}

SEXP Rf_getAttrib(SEXP s, SEXP tag)
{
	for (int i = 0; i < s->nattr; i++)
		if (s->attr_tag[i] == tag)
			return s->attr_val[i];
	return R_NilValue;
}

SEXP R_do_slot(SEXP s, SEXP tag) { return Rf_getAttrib(s, tag); }

SEXP Rf_allocMatrix(SEXPTYPE t, int nr, int nc)
{
	SEXP s = Rf_allocVector(t, (R_xlen_t) nr * nc);
	SEXP d = Rf_allocVector(INTSXP, 2);
	INTEGER(d)[0] = nr; INTEGER(d)[1] = nc;
	Rf_setAttrib(s, R_DimSymbol, d);
	return s;
}

SEXP Rf_allocArray(SEXPTYPE t, SEXP dim)
{
	R_xlen_t n = 1;
	for (int i = 0; i < LENGTH(dim); i++) n *= INTEGER(dim)[i];
	SEXP s = Rf_allocVector(t, n);
	Rf_setAttrib(s, R_DimSymbol, Rf_duplicate(dim));
	return s;
}

SEXP Rf_protect(SEXP s)
{
	if (++protect_depth > protect_max) protect_max = protect_depth;
	return s;
}

void Rf_unprotect(int n)
{
	protect_depth -= n;
	if (protect_depth < 0) { protect_underflow = 1; protect_depth = 0; }
}

SEXP Rf_duplicate(SEXP s)
{
	if (s == R_NilValue || s->type == SYMSXP || s->type == CHARSXP)
		return s;
	SEXP d = Rf_allocVector((SEXPTYPE) s->type, s->len);
	if (s->type == VECSXP)
		for (R_xlen_t i = 0; i < s->len; i++) SET_VECTOR_ELT(d, i, Rf_duplicate(VECTOR_ELT(s, i)));
	else
		memcpy(d->data, s->data, (size_t) s->len * elt_size((SEXPTYPE) s->type));
	for (int i = 0; i < s->nattr; i++)
		Rf_setAttrib(d, s->attr_tag[i], Rf_duplicate(s->attr_val[i]));
	return d;
}

SEXP Rf_mkChar(const char *str)
{
	SEXP s = new_rec(CHARSXP, (R_xlen_t) strlen(str), 1);
	strcpy((char *) s->data, str);
	return s;
}

SEXP Rf_install(const char *name)
{
	for (size_t i = 0; i < sym_n; i++)
		if (strcmp((const char *) symtab[i]->data, name) == 0)
			return symtab[i];
	SEXP s = (SEXP) xmalloc(sizeof(struct SEXPREC));          /* symbols live for the process */
	s->type = SYMSXP;
	s->len = (R_xlen_t) strlen(name);
	s->data = xmalloc(strlen(name) + 1);
	strcpy((char *) s->data, name);
	if (sym_n == sym_cap) {
		sym_cap = sym_cap ? sym_cap * 2 : 32;
		symtab = (SEXP *) realloc(symtab, sym_cap * sizeof(SEXP));
	}
	symtab[sym_n++] = s;
	return s;
}

SEXP Rf_mkString(const char *str)
{
	SEXP s = Rf_allocVector(STRSXP, 1);
	SET_STRING_ELT(s, 0, Rf_mkChar(str));
	return s;
}
SEXP Rf_ScalarInteger(int x) { SEXP s = Rf_allocVector(INTSXP, 1); INTEGER(s)[0] = x; return s; }
SEXP Rf_ScalarLogical(int x) { SEXP s = Rf_allocVector(LGLSXP, 1); LOGICAL(s)[0] = x; return s; }
SEXP Rf_ScalarReal(double x) { SEXP s = Rf_allocVector(REALSXP, 1); REAL(s)[0] = x; return s; }
SEXP Rf_ScalarString(SEXP c) { SEXP s = Rf_allocVector(STRSXP, 1); SET_STRING_ELT(s, 0, c); return s; }

SEXP Rf_coerceVector(SEXP s, SEXPTYPE t)
{
	if ((SEXPTYPE) s->type == t)
		return s;
	SEXP d = Rf_allocVector(t, s->len);
	for (R_xlen_t i = 0; i < s->len; i++) {
		if ((s->type == INTSXP || s->type == LGLSXP) && t == REALSXP)
			REAL(d)[i] = INTEGER(s)[i] == R_NaInt ? R_NaReal : (double) INTEGER(s)[i];
		else if (s->type == REALSXP && (t == INTSXP || t == LGLSXP))
			INTEGER(d)[i] = REAL(s)[i] != REAL(s)[i] ? R_NaInt : (int) REAL(s)[i];
		else if ((s->type == INTSXP || s->type == LGLSXP) && (t == INTSXP || t == LGLSXP))
			INTEGER(d)[i] = INTEGER(s)[i];
		else { fprintf(stderr, "r_standin: coerceVector %d -> %u not provided\n", s->type, t); abort(); }
	}
	for (int i = 0; i < s->nattr; i++)
		Rf_setAttrib(d, s->attr_tag[i], s->attr_val[i]);
	return d;
}

Rboolean Rf_isVectorList(SEXP s) { return s->type == VECSXP; }
Rboolean Rf_isNull(SEXP s) { return s == R_NilValue || s->type == NILSXP; }
Rboolean Rf_isInteger(SEXP s) { return s->type == INTSXP; }
Rboolean Rf_isLogical(SEXP s) { return s->type == LGLSXP; }
Rboolean Rf_isReal(SEXP s) { return s->type == REALSXP; }
Rboolean Rf_isNumeric(SEXP s) { return s->type == REALSXP || s->type == INTSXP || s->type == LGLSXP; }
Rboolean Rf_isString(SEXP s) { return s->type == STRSXP; }
Rboolean Rf_isBlankString(const char *s)
{
	for (; *s; s++)
		if (*s != ' ' && *s != '\t' && *s != '\n') return FALSE;
	return TRUE;
}
Rboolean Rf_isMatrix(SEXP s) { SEXP d = Rf_getAttrib(s, R_DimSymbol); return d != R_NilValue && LENGTH(d) == 2; }

static const struct { const char *name; SEXPTYPE t; } type_names[] = {
	{ "NULL", NILSXP }, { "logical", LGLSXP }, { "integer", INTSXP }, { "double", REALSXP },
	{ "numeric", REALSXP }, { "complex", CPLXSXP }, { "character", STRSXP }, { "list", VECSXP },
	{ "raw", RAWSXP }, { "symbol", SYMSXP },
};

SEXPTYPE Rf_str2type(const char *str)
{
	for (size_t i = 0; i < sizeof(type_names) / sizeof(type_names[0]); i++)
		if (strcmp(type_names[i].name, str) == 0)
			return type_names[i].t;
	return (SEXPTYPE) -1;
}

const char *Rf_type2char(SEXPTYPE t)
{
	for (size_t i = 0; i < sizeof(type_names) / sizeof(type_names[0]); i++)
		if (type_names[i].t == t)
			return type_names[i].name;
	return "unknown";
}

int Rf_asInteger(SEXP s)
{
	if (s->len < 1) return R_NaInt;
	if (s->type == REALSXP) return REAL(s)[0] != REAL(s)[0] ? R_NaInt : (int) REAL(s)[0];
	return INTEGER(s)[0];
}
int Rf_asLogical(SEXP s) { return Rf_asInteger(s); }
double Rf_asReal(SEXP s)
{
	if (s->len < 1) return R_NaReal;
	if (s->type == REALSXP) return REAL(s)[0];
	return INTEGER(s)[0] == R_NaInt ? R_NaReal : (double) INTEGER(s)[0];
}

char *R_alloc(size_t n, int size)
{
	if (arena_n == arena_cap) {
		arena_cap = arena_cap ? arena_cap * 2 : 64;
		arena = (void **) realloc(arena, arena_cap * sizeof(void *));
	}
	/* 64 guard bytes behind every block, checked when the call ends: a table that is too small shows */
	size_t bytes = n * (size_t) size;
	char *p = (char *) xmalloc(bytes + 64 + sizeof(size_t));
	*(size_t *) p = bytes;
	memset(p + sizeof(size_t) + bytes, 0xA5, 64);
	arena[arena_n++] = p;
	return p + sizeof(size_t);
}

void Rf_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(err_msg, sizeof(err_msg), fmt, ap);
	va_end(ap);
	if (err_jmp == NULL) { fprintf(stderr, "r_standin: error() outside a call: %s\n", err_msg); abort(); }
	longjmp(*err_jmp, 1);
}

void Rf_warning(const char *fmt, ...)
{
	char one[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(one, sizeof(one), fmt, ap);
	va_end(ap);
	size_t have = strlen(warn_log);
	snprintf(warn_log + have, sizeof(warn_log) - have, "%s%s", nwarn ? "\x1e" : "", one);
	nwarn++;
}

void R_CheckUserInterrupt(void) {}

/* ---- harness -------------------------------------------------------------------------------- */
SEXP sx_nil(void) { init_once(); return R_NilValue; }
SEXP sx_alloc(int type, long n) { return Rf_allocVector((SEXPTYPE) type, (R_xlen_t) n); }
void *sx_data(SEXP s) { return s->data; }
long sx_len(SEXP s) { return (long) s->len; }
int sx_type(SEXP s) { return s->type; }
SEXP sx_symbol(const char *name) { init_once(); return Rf_install(name); }
SEXP sx_mkchar(const char *str) { init_once(); return Rf_mkChar(str); }
SEXP sx_na_string(void) { init_once(); return R_NaString; }
const char *sx_char(SEXP s) { return (const char *) s->data; }
int sx_nattr(SEXP s) { return s->nattr; }
const char *sx_attr_name(SEXP s, int i) { return (const char *) s->attr_tag[i]->data; }
SEXP sx_attr_value(SEXP s, int i) { return s->attr_val[i]; }

typedef SEXP (*fn0)(void);
typedef SEXP (*fn1)(SEXP);
typedef SEXP (*fn2)(SEXP, SEXP);
typedef SEXP (*fn3)(SEXP, SEXP, SEXP);
typedef SEXP (*fn4)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn5)(SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn8)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn9)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);

/* One .Call: returns 0 and *out = the result, or 1 after error() (message: sx_error()).  The
   protection depth the call leaves behind, whether it ever popped more than it pushed, the warnings
   and the state of the R_alloc() guard bytes are read afterwards. */
static int guards_broken;
int sx_call(void *f, int nargs, SEXP *a, SEXP *out)
{
	init_once();
	jmp_buf jb;
	protect_depth = protect_max = protect_underflow = 0;
	warn_log[0] = 0; nwarn = 0; err_msg[0] = 0; guards_broken = 0;
	*out = R_NilValue;
	err_jmp = &jb;
	int failed = setjmp(jb);
	if (!failed) {
		switch (nargs) {
		case 0: *out = ((fn0) f)(); break;
		case 1: *out = ((fn1) f)(a[0]); break;
		case 2: *out = ((fn2) f)(a[0], a[1]); break;
		case 3: *out = ((fn3) f)(a[0], a[1], a[2]); break;
		case 4: *out = ((fn4) f)(a[0], a[1], a[2], a[3]); break;
		case 5: *out = ((fn5) f)(a[0], a[1], a[2], a[3], a[4]); break;
		case 6: *out = ((fn6) f)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
		case 7: *out = ((fn7) f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
		case 8: *out = ((fn8) f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
		case 9: *out = ((fn9) f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
		default: snprintf(err_msg, sizeof(err_msg), "sx_call: %d arguments", nargs); failed = 1;
		}
	}
	err_jmp = NULL;
	/* R releases R_alloc() memory when the .Call returns (or unwinds) */
	for (size_t i = 0; i < arena_n; i++) {
		char *p = (char *) arena[i];
		size_t bytes = *(size_t *) p;
		for (int g = 0; g < 64; g++)
			if ((unsigned char) p[sizeof(size_t) + bytes + g] != 0xA5) guards_broken++;
		free(p);
	}
	arena_n = 0;
	return failed;
}

const char *sx_error(void) { return err_msg; }
const char *sx_warnings(void) { return warn_log; }
int sx_nwarnings(void) { return nwarn; }
int sx_protect_depth(void) { return protect_depth; }
int sx_protect_max(void) { return protect_max; }
int sx_protect_underflow(void) { return protect_underflow; }
int sx_guards_broken(void) { return guards_broken; }

/* frees every SEXP made so far (symbols and the constants stay) */
void sx_reset(void)
{
	for (size_t i = 0; i < objs_n; i++) {
		if (objs[i] == R_NaString || objs[i] == R_BlankString)
			continue;
		free(objs[i]->data);
		free(objs[i]);
	}
	size_t keep = 0;
	if (R_NaString) objs[keep++] = R_NaString;
	if (R_BlankString) objs[keep++] = R_BlankString;
	objs_n = keep;
}
