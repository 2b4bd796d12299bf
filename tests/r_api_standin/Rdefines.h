/*
 * Test-only stand-in for R's <Rdefines.h>: DECLARATIONS of the part of R's public C API that
 * integration/svt_hip_glue.c and the headers it includes name, so that
 *     gcc -fsyntax-only -Wall -I tests/r_api_standin -I <reference>/src -I include integration/svt_hip_glue.c
 * (tests/test_glue_compiles.py) can check the glue's syntax, types, arities and PROTECT discipline
 * against a compiler, and so that tests/test_glue_executes.py can RUN the glue on the CPU against the
 * test-only definitions of r_standin.c (same directory).  Nothing here is shipped or linked into the
 * product: it exists because the image has no R.  Written from R's documented API ("Writing R Extensions").
 */
#ifndef SVT_TEST_RDEFINES_STANDIN_H
#define SVT_TEST_RDEFINES_STANDIN_H

#include <stddef.h>
#include <limits.h>
#include <string.h>     /* R's own headers pull it in; the reference's files rely on that */

typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int R_len_t;
typedef unsigned char Rbyte;
typedef enum { FALSE = 0, TRUE } Rboolean;
typedef union { struct { double r, i; }; double private_data_c[2]; } Rcomplex;
typedef unsigned int SEXPTYPE;

#define NILSXP 0
#define SYMSXP 1
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define CPLXSXP 15
#define STRSXP 16
#define VECSXP 19
#define RAWSXP 24

extern SEXP R_NilValue, R_NaString, R_BlankString, R_DimSymbol, R_DimNamesSymbol, R_NamesSymbol,
	    R_ClassSymbol;
extern double R_NaReal, R_NaN, R_PosInf, R_NegInf;
extern int R_NaInt;
#define NA_INTEGER R_NaInt
#define NA_LOGICAL R_NaInt
#define NA_REAL R_NaReal
#define NA_STRING R_NaString

int R_IsNA(double);
int R_IsNaN(double);
int R_finite(double);
#define ISNAN(x) ((x) != (x))
#define R_FINITE(x) R_finite(x)

int *INTEGER(SEXP);
int *LOGICAL(SEXP);
double *REAL(SEXP);
Rcomplex *COMPLEX(SEXP);
Rbyte *RAW(SEXP);
void *DATAPTR(SEXP);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
int LENGTH(SEXP);
R_xlen_t XLENGTH(SEXP);
int TYPEOF(SEXP);
SEXP ATTRIB(SEXP);
const char *CHAR(SEXP);

SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_allocArray(SEXPTYPE, SEXP);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
SEXP Rf_duplicate(SEXP);
SEXP Rf_install(const char *);
SEXP Rf_mkChar(const char *);
SEXP Rf_mkString(const char *);
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarLogical(int);
SEXP Rf_ScalarReal(double);
SEXP Rf_ScalarString(SEXP);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_coerceVector(SEXP, SEXPTYPE);
SEXP R_do_slot(SEXP, SEXP);
Rboolean Rf_isVectorList(SEXP);
Rboolean Rf_isNull(SEXP);
Rboolean Rf_isInteger(SEXP);
Rboolean Rf_isLogical(SEXP);
Rboolean Rf_isReal(SEXP);
Rboolean Rf_isNumeric(SEXP);
Rboolean Rf_isString(SEXP);
Rboolean Rf_isMatrix(SEXP);
Rboolean Rf_isBlankString(const char *);
SEXPTYPE Rf_str2type(const char *);
const char *Rf_type2char(SEXPTYPE);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
double Rf_asReal(SEXP);
char *R_alloc(size_t, int);
void Rf_error(const char *, ...) __attribute__((noreturn, format(printf, 1, 2)));
void Rf_warning(const char *, ...) __attribute__((format(printf, 1, 2)));
void R_CheckUserInterrupt(void);

#define allocVector Rf_allocVector
#define allocMatrix Rf_allocMatrix
#define allocArray Rf_allocArray
#define duplicate Rf_duplicate
#define install Rf_install
#define mkChar Rf_mkChar
#define mkString Rf_mkString
#define ScalarInteger Rf_ScalarInteger
#define ScalarLogical Rf_ScalarLogical
#define ScalarReal Rf_ScalarReal
#define ScalarString Rf_ScalarString
#define getAttrib Rf_getAttrib
#define setAttrib Rf_setAttrib
#define coerceVector Rf_coerceVector
#define isVectorList Rf_isVectorList
#define isNull Rf_isNull
#define isInteger Rf_isInteger
#define isLogical Rf_isLogical
#define isReal Rf_isReal
#define isNumeric Rf_isNumeric
#define isString Rf_isString
#define isMatrix Rf_isMatrix
#define isBlankString Rf_isBlankString
#define str2type Rf_str2type
#define type2char Rf_type2char
#define asInteger Rf_asInteger
#define asLogical Rf_asLogical
#define asReal Rf_asReal
#define error Rf_error
#define warning Rf_warning
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)

/* the Rdefines.h layer proper */
#define NEW_INTEGER(n) Rf_allocVector(INTSXP, n)
#define NEW_LOGICAL(n) Rf_allocVector(LGLSXP, n)
#define NEW_NUMERIC(n) Rf_allocVector(REALSXP, n)
#define NEW_CHARACTER(n) Rf_allocVector(STRSXP, n)
#define NEW_LIST(n) Rf_allocVector(VECSXP, n)
#define IS_INTEGER(x) Rf_isInteger(x)
#define IS_LOGICAL(x) Rf_isLogical(x)
#define IS_NUMERIC(x) Rf_isReal(x)
#define IS_CHARACTER(x) Rf_isString(x)
#define IS_LIST(x) Rf_isVectorList(x)
#define GET_LENGTH(x) LENGTH(x)
#define GET_SLOT(x, what) R_do_slot(x, what)
#define GET_DIM(x) Rf_getAttrib(x, R_DimSymbol)
#define GET_DIMNAMES(x) Rf_getAttrib(x, R_DimNamesSymbol)
#define GET_NAMES(x) Rf_getAttrib(x, R_NamesSymbol)
#define GET_CLASS(x) Rf_getAttrib(x, R_ClassSymbol)
#define SET_DIM(x, v) Rf_setAttrib(x, R_DimSymbol, v)
#define SET_DIMNAMES(x, v) Rf_setAttrib(x, R_DimNamesSymbol, v)
#define SET_NAMES(x, v) Rf_setAttrib(x, R_NamesSymbol, v)
#define AS_INTEGER(x) Rf_coerceVector(x, INTSXP)
#define AS_NUMERIC(x) Rf_coerceVector(x, REALSXP)

#endif
