#!/usr/bin/env python3
"""Generate tests/golden/golden.json -- the known-answer vectors that the
reference's own test-suite holds for the SVT hot path.

The reference tests (tests/testthat/*.R, man/*.Rd examples) all have the same
shape: build a small dense matrix/array, coerce it to SVT_SparseArray, run the
operation on the sparse object and require the result to be identical (or
all.equal) to *base R / matrixStats applied to the dense object*.  This script
re-types those dense inputs (file:line cited per block) and computes the
expected outputs with an independent, dense, pure-numpy statement of the base R
semantics (NA vs NaN rules included).  It does NOT use the oracle or the HIP
library, and it does not read the reference at run time.

Run:  python tests/golden/make_golden.py   (rewrites golden.json)
"""
from __future__ import annotations

import json
import math
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

NA_INT = -2 ** 31
NA_REAL = np.frombuffer(struct.pack("<Q", 0x7FF00000000007A2), dtype=np.float64)[0]
NAN = float("nan")
INF = float("inf")


def is_na_real(x):
    x = np.asarray(x, dtype=np.float64)
    return np.isnan(x) & ((x.view(np.uint64) & np.uint64(0xFFFFFFFF)) == np.uint64(1954))


# ---------------------------------------------------------------------------
# encoding
# ---------------------------------------------------------------------------
def enc(a):
    a = np.asarray(a)
    if a.dtype == np.bool_:
        a = a.astype(np.int32)
    if a.dtype.kind == "i":
        a = a.astype(np.int32)
        dt = "i4"
    else:
        a = a.astype(np.float64)
        dt = "f8"
    return {"__nd__": True, "dtype": dt, "shape": list(a.shape),
            "hex": np.asfortranarray(a).tobytes(order="F").hex()}


def svt(dense, type_, na_bg=False):
    """na_bg: an NaArray (as(dense, "NaArray")): same dense object, NA background."""
    o = {"__svt__": True, "type": type_, "dense": enc(dense)}
    if na_bg:
        o["na_background"] = True
    return o


CASES = []


def case(src, fn, args, expected, cmp="identical", kwargs=None, warn=None,
         error=None, note=None):
    c = {"id": len(CASES), "src": src, "fn": fn, "args": args,
         "kwargs": kwargs or {}, "cmp": cmp}
    if expected is not None:
        c["expected"] = enc(expected)
    if warn is not None:
        c["warn"] = warn
    if error is not None:
        c["error"] = error
    if note:
        c["note"] = note
    CASES.append(c)


# ---------------------------------------------------------------------------
# dense base-R semantics
# ---------------------------------------------------------------------------
def rmat(nrow, ncol, fill=0.0, dtype=np.float64):
    return np.full((nrow, ncol), fill, dtype=dtype, order="F")


def set_lin(a, idx1, vals):
    """a[idx] <- vals with 1-based column-major linear indices."""
    flat = a.reshape(-1, order="F").copy()
    for i, v in zip(idx1, vals):
        flat[i - 1] = v
    return flat.reshape(a.shape, order="F")


def int_to_double(a):
    out = a.astype(np.float64)
    out[a == NA_INT] = NA_REAL
    return out


def crossprod_dense(a, b):
    """t(a) %*% b with plain IEEE arithmetic, k ascending (dense: zeros take
    part, so 0*Inf = NaN).  Integer inputs: any NA factor makes the cell NA."""
    is_int = a.dtype == np.int32
    if is_int:
        a_na, b_na = a == NA_INT, b == NA_INT
        a, b = a.astype(np.float64), b.astype(np.float64)
    out = np.zeros((a.shape[1], b.shape[1]), dtype=np.float64, order="F")
    with np.errstate(all="ignore"):
        for i in range(a.shape[1]):
            for j in range(b.shape[1]):
                if is_int and (a_na[:, i].any() or b_na[:, j].any()):
                    out[i, j] = NA_REAL
                    continue
                acc = 0.0
                has_na = False
                for k in range(a.shape[0]):
                    if is_na_real(a[k, i]) or is_na_real(b[k, j]):
                        has_na = True
                    acc += a[k, i] * b[k, j]
                out[i, j] = NA_REAL if has_na else acc
    return out


def col_apply(a, f):
    return np.array([f(a[:, j]) for j in range(a.shape[1])])


def vals_d(v, narm):
    v = np.asarray(v, dtype=np.float64)
    return v[~np.isnan(v)] if narm else v


def vals_i(v, narm):
    v = np.asarray(v, dtype=np.int32)
    return v[v != NA_INT] if narm else v


def has_na(v):
    v = np.asarray(v)
    if v.dtype == np.int32:
        return bool((v == NA_INT).any())
    return bool(is_na_real(v).any())


def has_nan_any(v):
    v = np.asarray(v)
    if v.dtype == np.int32:
        return bool((v == NA_INT).any())
    return bool(np.isnan(v).any())


def asd(v):
    v = np.asarray(v)
    return int_to_double(v) if v.dtype == np.int32 else v


def r_sum(v, narm=False):
    v = vals_i(v, narm) if np.asarray(v).dtype == np.int32 else vals_d(v, narm)
    if has_na(v):
        return NA_REAL
    d = asd(v)
    if np.isnan(d).any():
        return NAN
    with np.errstate(all="ignore"):
        return float(np.sum(d.astype(np.longdouble)))


def r_prod(v, narm=False):
    v = vals_i(v, narm) if np.asarray(v).dtype == np.int32 else vals_d(v, narm)
    if has_na(v):
        return NA_REAL
    d = asd(v)
    if np.isnan(d).any():
        return NAN
    p = 1.0
    with np.errstate(all="ignore"):
        for x in d:
            p *= x
    return p


def r_mean(v, narm=False):
    v = vals_i(v, narm) if np.asarray(v).dtype == np.int32 else vals_d(v, narm)
    n = len(v)
    s = r_sum(v)
    if np.isnan(s):
        return s
    if n == 0:
        return NAN
    return s / n


def r_var(v, narm=False):
    v = vals_i(v, narm) if np.asarray(v).dtype == np.int32 else vals_d(v, narm)
    n = len(v)
    if n <= 1:
        return NA_REAL
    if has_na(v):
        return NA_REAL
    d = asd(v)
    if np.isnan(d).any():
        return NAN
    with np.errstate(all="ignore"):
        m = float(np.sum(d.astype(np.longdouble))) / n
        return float(np.sum((d - m) ** 2)) / (n - 1)


def r_sd(v, narm=False):
    x = r_var(v, narm)
    with np.errstate(all="ignore"):
        return x if np.isnan(x) else math.sqrt(x)


def r_minmax_d(v, narm, is_min):
    v = vals_d(v, narm)
    if has_na(v):
        return NA_REAL
    if np.isnan(v).any():
        return NAN
    if len(v) == 0:
        return INF if is_min else -INF
    return float(v.min() if is_min else v.max())


def r_minmax_i(v, narm, is_min):
    """int min/max; empty / all-NA-with-na.rm gives NA (the documented
    deviation, src/Rvector_summarization.c:1108-1128)."""
    v = vals_i(v, narm)
    if has_na(v) or len(v) == 0:
        return NA_INT
    return int(v.min() if is_min else v.max())


def r_minmax(v, narm, is_min):
    v = np.asarray(v)
    if v.dtype == np.int32:
        return r_minmax_i(v, narm, is_min)
    return r_minmax_d(v, narm, is_min)


def r_any(v, narm=False):
    v = vals_i(v, narm)
    if ((v != 0) & (v != NA_INT)).any():
        return 1
    return NA_INT if has_na(v) else 0


def r_all(v, narm=False):
    v = vals_i(v, narm)
    if (v == 0).any():
        return 0
    return NA_INT if has_na(v) else 1


def r_anyNA(v):
    return 1 if has_nan_any(v) else 0


def slices_col(a, dims):
    """Generalized columns: one flat vector per position of tail(dim,-dims)."""
    inner = int(np.prod(a.shape[:dims], dtype=np.int64))
    outer = int(np.prod(a.shape[dims:], dtype=np.int64))
    flat = a.reshape((inner, outer), order="F")
    out_shape = a.shape[dims:]
    return [flat[:, g] for g in range(outer)], out_shape


def slices_row(a, dims):
    inner = int(np.prod(a.shape[:dims], dtype=np.int64))
    outer = int(np.prod(a.shape[dims:], dtype=np.int64))
    flat = a.reshape((inner, outer), order="F")
    return [flat[i, :] for i in range(inner)], a.shape[:dims]


def stat_col(a, f, dims=1, dtype=np.float64):
    sl, shp = slices_col(a, dims)
    out = np.array([f(s) for s in sl], dtype=dtype)
    return out.reshape(shp, order="F") if len(shp) > 1 else out


def stat_row(a, f, dims=1, dtype=np.float64):
    sl, shp = slices_row(a, dims)
    out = np.array([f(s) for s in sl], dtype=dtype)
    return out.reshape(shp, order="F") if len(shp) > 1 else out


def fix_sym(cp):
    """.fix_sym_mat_NA_NaN_pattern, tests/testthat/test-SparseMatrix-mult.R:7-17"""
    cp = cp.copy()
    isnan_only = np.isnan(cp) & ~is_na_real(cp)
    not_sym = isnan_only != isnan_only.T
    cp[not_sym] = NA_REAL
    return cp


# ---------------------------------------------------------------------------
# A. crossprod / tcrossprod, type "double"
#    tests/testthat/test-SparseMatrix-mult.R:206-272
# ---------------------------------------------------------------------------
SRC = "tests/testthat/test-SparseMatrix-mult.R"


def sym_crossprod_cases(m, tag, src):
    t = "double" if m.dtype == np.float64 else "integer"
    cmp = "equal" if t == "double" else "identical"
    cp = crossprod_dense(m, m)
    if t == "double":
        cp = fix_sym(cp)
    case(src, "crossprod", [svt(m, t)], cp, cmp, note=tag + " crossprod(svt)")
    case(src, "crossprod", [svt(m, t), svt(m, t)], cp, cmp)
    case(src, "crossprod", [svt(m, t), enc(m)], cp, cmp)
    case(src, "crossprod", [enc(m), svt(m, t)], cp, cmp)
    tm = np.asfortranarray(m.T)
    tcp = crossprod_dense(tm, tm)
    if t == "double":
        tcp = fix_sym(tcp)
    case(src, "tcrossprod", [svt(m, t)], tcp, cmp, note=tag + " tcrossprod(svt)")
    case(src, "tcrossprod", [svt(m, t), svt(m, t)], tcp, cmp)
    case(src, "tcrossprod", [svt(m, t), enc(m)], tcp, cmp)
    case(src, "tcrossprod", [enc(m), svt(m, t)], tcp, cmp)


m0 = rmat(5, 3)
m0[2, 0] = INF
m0[1, 2] = -11.99
sym_crossprod_cases(m0, "m0", SRC + ":207-211")

m1 = np.array([[0, -4.5, 7, NA_REAL, 0, NAN, INF, -INF]], dtype=np.float64, order="F")
sym_crossprod_cases(m1, "m1", SRC + ":213-215")

m2 = set_lin(rmat(6, 4), [24, 1, 2, 8, 10, 15, 16, 17], [k - 3.5 for k in range(1, 9)])
sym_crossprod_cases(m2, "m2", SRC + ":217-220")

m3 = rmat(6, 7)
m3 = set_lin(m3, [3 + 4 * k for k in range(10)], [2.4 ** k for k in range(1, 11)])
m3 = set_lin(m3, [4 + 4 * k for k in range(10)], [-(101 + k) for k in range(10)])
m3[0, 4] = NAN
m3[4, 2] = INF
sym_crossprod_cases(m3, "m3", SRC + ":222-229")

exp23 = crossprod_dense(m2, m3)
S2, S3 = svt(m2, "double"), svt(m3, "double")
case(SRC + ":231-236", "crossprod", [S2, S3], exp23, "identical")
case(SRC + ":231-236", "crossprod", [S2, enc(m3)], exp23, "identical")
case(SRC + ":231-236", "crossprod", [enc(m2), S3], exp23, "identical")
case(SRC + ":231-236", "crossprod", [S3, S2], np.asfortranarray(exp23.T), "identical")
tm2, tm3 = np.asfortranarray(m2.T), np.asfortranarray(m3.T)
case(SRC + ":237-246", "tcrossprod", [svt(tm2, "double"), svt(tm3, "double")], exp23, "identical")
case(SRC + ":237-246", "tcrossprod", [svt(tm2, "double"), enc(tm3)], exp23, "identical")
case(SRC + ":237-246", "tcrossprod", [enc(tm2), svt(tm3, "double")], exp23, "identical")
case(SRC + ":237-246", "tcrossprod", [svt(tm3, "double"), svt(tm2, "double")],
     np.asfortranarray(exp23.T), "identical")

m4 = rmat(0, 3)
sym_crossprod_cases(m4, "m4 zero rows", SRC + ":248-251")
m5 = rmat(6, 0)
sym_crossprod_cases(m5, "m5 zero cols", SRC + ":253-256")
exp35 = crossprod_dense(m3, m5)
case(SRC + ":258-271", "crossprod", [S3, svt(m5, "double")], exp35)
case(SRC + ":258-271", "crossprod", [S3, enc(m5)], exp35)
case(SRC + ":258-271", "crossprod", [enc(m3), svt(m5, "double")], exp35)
case(SRC + ":258-271", "crossprod", [svt(m5, "double"), S3], np.asfortranarray(exp35.T))
tm5 = np.asfortranarray(m5.T)
case(SRC + ":258-271", "tcrossprod", [svt(tm3, "double"), svt(tm5, "double")], exp35)
case(SRC + ":258-271", "tcrossprod", [svt(tm3, "double"), enc(tm5)], exp35)
case(SRC + ":258-271", "tcrossprod", [enc(tm3), svt(tm5, "double")], exp35)
case(SRC + ":258-271", "tcrossprod", [svt(tm5, "double"), svt(tm3, "double")],
     np.asfortranarray(exp35.T))

# ---------------------------------------------------------------------------
# B. crossprod, type "integer"  (:279-304 with the helper at :65-199)
# ---------------------------------------------------------------------------


def int_crossprod_cases(a, b, src):
    def variants(m):
        d = int_to_double(m)
        return [svt(m, "integer"), svt(d, "double")], [enc(m)]

    def self_cases(m):
        cp = crossprod_dense(m, m)
        (si, sd), (mi,) = variants(m)
        case(src, "crossprod", [si], cp)
        case(src, "crossprod", [sd], cp)
        for x in (si, sd):
            for y in (si, sd, mi):
                case(src, "crossprod", [x, y], cp)
        case(src, "crossprod", [mi, si], cp)
        case(src, "crossprod", [mi, sd], cp)

    self_cases(a)
    if b is None:
        return
    self_cases(b)
    exp = crossprod_dense(a, b)
    (ai, ad), (am,) = variants(a)
    (bi, bd), (bm,) = variants(b)
    for x in (ai, ad):
        for y in (bi, bd, bm):
            case(src, "crossprod", [x, y], exp)
    case(src, "crossprod", [am, bi], exp)
    case(src, "crossprod", [am, bd], exp)
    expt = np.asfortranarray(exp.T)
    for x in (bi, bd):
        for y in (ai, ad, am):
            case(src, "crossprod", [x, y], expt)
    case(src, "crossprod", [bm, ai], expt)
    case(src, "crossprod", [bm, ad], expt)


im1 = np.array([[0, -4, 7, NA_INT, 0, NA_INT]], dtype=np.int32, order="F")
int_crossprod_cases(im1, None, SRC + ":280-281")
im2 = set_lin(rmat(6, 4, 0, np.int32), [24, 1, 2, 8, 10, 15, 16, 17],
              [k * 10 - 35 for k in range(1, 9)])
im3 = rmat(6, 7, 0, np.int32)
im3 = set_lin(im3, [3 + 4 * k for k in range(10)], list(range(1, 11)))
im3 = set_lin(im3, [4 + 4 * k for k in range(10)], [-(101 + k) for k in range(10)])
int_crossprod_cases(im2, im3, SRC + ":283-290")
im2n, im3n = im2.copy(), im3.copy()
im2n[1, 3] = NA_INT
im3n[0, 4] = NA_INT
int_crossprod_cases(im2n, im3n, SRC + ":292-294")
im4 = rmat(0, 3, 0, np.int32)
int_crossprod_cases(im4, None, SRC + ":296-298")
im5 = rmat(6, 0, 0, np.int32)
int_crossprod_cases(im5, None, SRC + ":300-302")
int_crossprod_cases(im3, im5, SRC + ":303")

# ---------------------------------------------------------------------------
# C. %*%   (:306-321; man/SparseMatrix-mult.Rd:63-92)
#    The reference draws m2 with runif(12) under set.seed(333); R's RNG is not
#    available here, so a fixed 6x2 double matrix stands in for it.
# ---------------------------------------------------------------------------
mm1 = set_lin(rmat(15, 6, 0, np.int32),
              [2, 6] + list(range(12, 18)) + list(range(22, 34)) + [55] +
              list(range(59, 63)) + [90], list(range(101, 127)))
mm2 = np.array([[0.46728, 0.08459], [0.83970, 0.72096], [0.34608, 0.10771],
                [0.57142, 0.39572], [0.02011, 0.22865], [0.72355, 0.69431]],
               dtype=np.float64, order="F")
mm1d = mm1.astype(np.float64)
exp_mm = crossprod_dense(np.asfortranarray(mm1d.T), mm2)
case(SRC + ":306-321", "matmul", [svt(mm1, "integer"), svt(mm2, "double")], exp_mm, "identical")
case(SRC + ":306-321", "matmul", [svt(mm1, "integer"), enc(mm2)], exp_mm, "identical")
case(SRC + ":306-321", "matmul", [enc(mm1), svt(mm2, "double")], exp_mm, "identical")
case("man/SparseMatrix-mult.Rd:78-91", "crossprod", [svt(mm1, "integer")],
     crossprod_dense(mm1, mm1))
case("man/SparseMatrix-mult.Rd:78-91", "tcrossprod", [svt(mm1, "integer")],
     crossprod_dense(np.asfortranarray(mm1.T), np.asfortranarray(mm1.T)))

# ---------------------------------------------------------------------------
# D. matrixStats, 2-D integer + logical
#    tests/testthat/test-SparseArray-matrixStats.R:141-242
# ---------------------------------------------------------------------------
SRC_MS = "tests/testthat/test-SparseArray-matrixStats.R"
NA = NA_INT
ms1 = np.array([[0, 0, NA, 0, NA],
                [NA, 0, -3, 1, NA],
                [0, 0, 0, 0, 0],
                [15, 0, 0, 0, NA]], dtype=np.int32, order="F")
ms2 = (ms1 == NA_INT).astype(np.int32)   # is.na(m1), logical


def matrixstats_2d(m, t, src, with_anyall_pinned, na_bg=False):
    """na_bg: the same cases on as(m, "NaArray") (tests/testthat/test-NaArray-matrixStats.R:
    expected = the same base-R results on the same dense matrix); col* only, the
    row* statistics of NaArray objects are not implemented yet."""
    if na_bg:
        emitted = len(CASES)
    S = svt(m, t, na_bg)
    tm = np.asfortranarray(m.T)
    for narm in (False, True):
        kw = {"na_rm": narm}
        anyall_note = None if with_anyall_pinned else \
            "integer any/all: (x != 0) as in src/Rvector_summarization.c:270-322"
        case(src, "colAnys", [S], stat_col(m, lambda v: r_any(v, narm), dtype=np.int32), kwargs=kw, note=anyall_note)
        case(src, "rowAnys", [S], stat_col(tm, lambda v: r_any(v, narm), dtype=np.int32), kwargs=kw, note=anyall_note)
        case(src, "colAlls", [S], stat_col(m, lambda v: r_all(v, narm), dtype=np.int32), kwargs=kw, note=anyall_note)
        case(src, "rowAlls", [S], stat_col(tm, lambda v: r_all(v, narm), dtype=np.int32), kwargs=kw, note=anyall_note)
        for nm, is_min in (("Mins", True), ("Maxs", False)):
            case(src, "col" + nm, [S], stat_col(m, lambda v: r_minmax_i(v, narm, is_min), dtype=np.int32), kwargs=kw)
            case(src, "row" + nm, [S], stat_col(tm, lambda v: r_minmax_i(v, narm, is_min), dtype=np.int32), kwargs=kw)
        for pre, mm in (("col", m), ("row", tm)):
            lo = stat_col(mm, lambda v: r_minmax_i(v, narm, True), dtype=np.int32)
            hi = stat_col(mm, lambda v: r_minmax_i(v, narm, False), dtype=np.int32)
            case(src, pre + "Ranges", [S], np.stack([lo, hi], axis=-1), kwargs=kw)
            case(src, pre + "Sums", [S], stat_col(mm, lambda v: r_sum(v, narm)), kwargs=kw)
            case(src, pre + "Sums2", [S], stat_col(mm, lambda v: r_sum(v, narm)), kwargs=kw)
            case(src, pre + "Prods", [S], stat_col(mm, lambda v: r_prod(v, narm)), kwargs=kw)
            case(src, pre + "Means", [S], stat_col(mm, lambda v: r_mean(v, narm)), kwargs=kw)
            case(src, pre + "Means2", [S], stat_col(mm, lambda v: r_mean(v, narm)), kwargs=kw)
            case(src, pre + "Vars", [S], stat_col(mm, lambda v: r_var(v, narm)), "equal", kwargs=kw)
            case(src, pre + "Sds", [S], stat_col(mm, lambda v: r_sd(v, narm)), "equal", kwargs=kw)
    # zero-row object: NA + warning for the col ops (:173-192)
    z = m[0:0, :]
    Z = svt(z, t, na_bg)
    na5 = np.full(5, NA_INT, dtype=np.int32)
    case(src, "colMins", [Z], na5, warn="NAs introduced")
    case(src, "colMaxs", [Z], na5, warn="NAs introduced")
    case(src, "colRanges", [Z], np.stack([na5, na5], axis=-1), warn="NAs introduced")
    case(src, "rowMins", [Z], np.zeros(0, np.int32))
    case(src, "rowMaxs", [Z], np.zeros(0, np.int32))
    if na_bg:
        # row* methods the reference defines for NaArray objects (R/NaArray-matrixStats.R;
        # rowAnys/Alls/Prods/Means/Vars/Sds are commented out there)
        ok_row = {"rowMins", "rowMaxs", "rowRanges", "rowSums", "rowSums2", "rowAnyNAs"}
        CASES[emitted:] = [c for c in CASES[emitted:] if not c["fn"].startswith("row") or c["fn"] in ok_row]
        for i, c in enumerate(CASES):
            c["id"] = i


matrixstats_2d(ms1, "integer", SRC_MS + ":141-192", with_anyall_pinned=False)
matrixstats_2d(ms2, "logical", SRC_MS + ":194-242", with_anyall_pinned=True)
SRC_NM = "tests/testthat/test-NaArray-matrixStats.R"
matrixstats_2d(ms1, "integer", SRC_NM + ":130-178", with_anyall_pinned=False, na_bg=True)
matrixstats_2d(ms2, "logical", SRC_NM + ":180-228", with_anyall_pinned=True, na_bg=True)

# E. col/rowAnyNAs (:69-106)
an1 = np.array([[0, 0, 155], [0, 8, -1]], dtype=np.int32, order="F")
an1n = an1.copy(); an1n[0, 1] = NA_INT
an2 = np.array([[0, 0, 1], [0, 1, 1]], dtype=np.int32, order="F")
an2n = an2.copy(); an2n[0, 1] = NA_INT
an3 = np.array([[0, 0, math.pi], [0, 0.25, 1e3]], dtype=np.float64, order="F")
an3a = an3.copy(); an3a[0, 1] = NAN
an3b = an3.copy(); an3b[0, 1] = NA_REAL
for m, t in ((an1, "integer"), (an1n, "integer"), (an2, "logical"), (an2n, "logical"),
             (an3, "double"), (an3a, "double"), (an3b, "double")):
    case(SRC_MS + ":69-106", "colAnyNAs", [svt(m, t)], stat_col(m, r_anyNA, dtype=np.int32))
    case(SRC_MS + ":69-106", "rowAnyNAs", [svt(m, t)],
         stat_col(np.asfortranarray(m.T), r_anyNA, dtype=np.int32))
    case("tests/testthat/test-SparseArray-summarization.R:2-31", "anyNA", [svt(m, t)],
         np.int32(r_anyNA(m.reshape(-1))))
    case(SRC_NM + ":56-100", "colAnyNAs", [svt(m, t, True)], stat_col(m, r_anyNA, dtype=np.int32))
    case(SRC_NM + ":56-100", "rowAnyNAs", [svt(m, t, True)],
         stat_col(np.asfortranarray(m.T), r_anyNA, dtype=np.int32))
    case("tests/testthat/test-NaArray-summarization.R:1-31", "anyNA", [svt(m, t, True)],
         np.int32(r_anyNA(m.reshape(-1))))

# ---------------------------------------------------------------------------
# F. 3-D double  (:244-270)
# ---------------------------------------------------------------------------
a3 = np.zeros((6, 5, 4), dtype=np.float64, order="F")
a3[0, :, 1] = [1e12, -1234.55, -2.1, -1, -0.55]
a3[2, :, 1] = [-0.55, 0, 1e-10, 0.88, 1]
a3[4, :, 1] = [math.pi, 10.33, 3.4567895e8, 300, 2009.01]
a3_clean = a3.copy()
a3[5, 2, 1] = NA_REAL
a3[5, 3, 1] = NAN


def minmax_3d(a, t, src, warn_ok):
    """test_3D_colrowMinsMaxs, tests/testthat/helpers.R:262-320"""
    S = svt(a, t)
    dt = np.int32 if t != "double" else np.float64
    for narm in (False, True):
        for dims in (1, 2):
            for nm, is_min in (("Mins", True), ("Maxs", False)):
                kw = {"na_rm": narm, "dims": dims}
                case(src, "col" + nm, [S], stat_col(a, lambda v: r_minmax(v, narm, is_min), dims, dt),
                     kwargs=kw, warn="*" if warn_ok else None)
                case(src, "row" + nm, [S], stat_row(a, lambda v: r_minmax(v, narm, is_min), dims, dt),
                     kwargs=kw, warn="*" if warn_ok else None)


minmax_3d(a3, "double", SRC_MS + ":253-254", False)
S3d = svt(a3, "double")
for dims in (1, 2):
    for narm in (False, True):
        kw = {"na_rm": narm, "dims": dims}
        case(SRC_MS + ":256-268", "colSums", [S3d], stat_col(a3, lambda v: r_sum(v, narm), dims), "equal", kwargs=kw)
        case(SRC_MS + ":256-268", "rowSums", [S3d], stat_row(a3, lambda v: r_sum(v, narm), dims), "equal", kwargs=kw)
        case(SRC_MS + ":256-268", "colMeans", [S3d], stat_col(a3, lambda v: r_mean(v, narm), dims), "equal", kwargs=kw)
        case(SRC_MS + ":256-268", "rowMeans", [S3d], stat_row(a3, lambda v: r_mean(v, narm), dims), "equal", kwargs=kw)

# ---------------------------------------------------------------------------
# G. min/max torture  (:272-330)
# ---------------------------------------------------------------------------
t1 = np.array([[NA_INT, -8, 0], [0, 0, 1]], dtype=np.int32, order="F")
t2 = np.array([[0, NA_INT, 0, 0], [8, 9, 1, 1], [-8, -9, -10, -11]], dtype=np.int32, order="F")
for m in (t1, t2):
    S = svt(m, "integer")
    tm = np.asfortranarray(m.T)
    for narm in (False, True):
        for nm, is_min in (("Mins", True), ("Maxs", False)):
            kw = {"na_rm": narm}
            case(SRC_MS + ":276-288", "row" + nm, [S], stat_col(tm, lambda v: r_minmax_i(v, narm, is_min), dtype=np.int32), kwargs=kw)
            case(SRC_MS + ":276-288", "col" + nm, [S], stat_col(m, lambda v: r_minmax_i(v, narm, is_min), dtype=np.int32), kwargs=kw)

g0 = np.zeros((5, 4, 3), dtype=np.int32, order="F")
g0 = set_lin(g0, [1, 6, 16, 20, 21, 22, 36, 39, 40, 60],
             [2, -5, NA_INT, NA_INT, -11, 99, -8, NA_INT, NA_INT, NA_INT])
minmax_3d(g0, "integer", SRC_MS + ":293-303", True)
case(SRC_MS + ":301-302", "rowMins", [svt(g0, "integer")], None, kwargs={"na_rm": True, "dims": 2}, warn="NAs introduced")
case(SRC_MS + ":301-302", "rowMaxs", [svt(g0, "integer")], None, kwargs={"na_rm": True, "dims": 2}, warn="NAs introduced")
for sl, which in ((g0[:, :, 0:0], "k0"), (g0[:, 0:0, :], "j0"), (g0[0:0, :, :], "i0")):
    minmax_3d(np.asfortranarray(sl), "integer", SRC_MS + ":305-323 " + which, True)
z = svt(np.asfortranarray(g0[:, :, 0:0]), "integer")
for fn, kw in (("rowMins", {}), ("rowMaxs", {}), ("rowMins", {"dims": 2}), ("rowMaxs", {"dims": 2})):
    case(SRC_MS + ":305-310", fn, [z], None, kwargs=kw, warn="NAs introduced")
z = svt(np.asfortranarray(g0[:, 0:0, :]), "integer")
for fn, kw in (("rowMins", {}), ("rowMaxs", {}), ("colMins", {"dims": 2}), ("colMaxs", {"dims": 2})):
    case(SRC_MS + ":312-317", fn, [z], None, kwargs=kw, warn="NAs introduced")
z = svt(np.asfortranarray(g0[0:0, :, :]), "integer")
for fn, kw in (("colMins", {}), ("colMaxs", {}), ("colMins", {"dims": 2}), ("colMaxs", {"dims": 2})):
    case(SRC_MS + ":319-324", fn, [z], None, kwargs=kw, warn="NAs introduced")
g0d = int_to_double(g0)
g0d = set_lin(g0d, [39, 40], [NAN, NAN])
minmax_3d(g0d, "double", SRC_MS + ":326-329", False)
for sl, which in ((g0d[:, :, 0:0], "k0"), (g0d[:, 0:0, :], "j0"), (g0d[0:0, :, :], "i0")):
    minmax_3d(np.asfortranarray(sl), "double", SRC_MS + ":326-329 " + which, False)

# ---------------------------------------------------------------------------
# H. whole-array summarization
#    tests/testthat/test-SparseArray-summarization.R:56-126
# ---------------------------------------------------------------------------
SRC_SU = "tests/testthat/test-SparseArray-summarization.R"
INT_MAX = 2 ** 31 - 1


def int_or_double(v):
    if np.isnan(v):
        return np.int32(NA_INT)
    if -INT_MAX <= v <= INT_MAX:
        return np.int32(int(round(v)))
    return np.float64(v)


def summarize_cases(a, t, src, na_bg=False):
    S = svt(a, t, na_bg)
    v = a.reshape(-1, order="F")
    isint = t != "double"
    for narm in (False, True):
        kw = {"na_rm": narm}
        if isint:
            case(src, "any", [S], np.int32(r_any(v, narm)), kwargs=kw,
                 note=None if t == "logical" else "integer any/all: (x != 0)")
            case(src, "all", [S], np.int32(r_all(v, narm)), kwargs=kw,
                 note=None if t == "logical" else "integer any/all: (x != 0)")
        dt = np.int32 if isint else np.float64
        lo = r_minmax(v, narm, True)
        hi = r_minmax(v, narm, False)
        case(src, "min", [S], np.array(lo, dtype=dt), kwargs=kw)
        case(src, "max", [S], np.array(hi, dtype=dt), kwargs=kw)
        case(src, "range", [S], np.array([lo, hi], dtype=dt), kwargs=kw)
        s, p = r_sum(v, narm), r_prod(v, narm)
        if isint:
            case(src, "sum", [S], int_or_double(s), kwargs=kw)
            case(src, "prod", [S], int_or_double(p), kwargs=kw)
        else:
            case(src, "sum", [S], np.float64(s), "equal", kwargs=kw)
            case(src, "prod", [S], np.float64(p), "equal", kwargs=kw)
        case(src, "mean", [S], np.float64(r_mean(v, narm)), "equal" if not isint else "identical", kwargs=kw)
        case(src, "var", [S], np.float64(r_var(v, narm)), "equal", kwargs=kw)
        case(src, "sd", [S], np.float64(r_sd(v, narm)), "equal", kwargs=kw)
    if isint:
        Z = svt(a[0:0, :], t, na_bg)
        case(src, "min", [Z], np.int32(NA_INT), warn="NAs introduced")
        case(src, "max", [Z], np.int32(NA_INT), warn="NAs introduced")
        case(src, "range", [Z], np.array([NA_INT, NA_INT], np.int32), warn="NAs introduced")


summarize_cases(ms1, "integer", SRC_SU + ":56-79")
summarize_cases(ms2, "logical", SRC_SU + ":81-103")
case(SRC_SU + ":105-112", "anyNA", [svt(a3_clean, "double")], np.int32(0))
case(SRC_SU + ":113-115", "anyNA", [S3d], np.int32(1))
case(SRC_SU + ":116-117", "any", [S3d], None, error="does not support")
case(SRC_SU + ":116-117", "all", [S3d], None, error="does not support")
summarize_cases(a3, "double", SRC_SU + ":118-125")
SRC_NS = "tests/testthat/test-NaArray-summarization.R"
summarize_cases(ms1, "integer", SRC_NS + ":56-77", na_bg=True)
summarize_cases(ms2, "logical", SRC_NS + ":79-100", na_bg=True)
case(SRC_NS + ":104-110", "anyNA", [svt(a3_clean, "double", True)], np.int32(0))
case(SRC_NS + ":111-114", "anyNA", [svt(a3, "double", True)], np.int32(1))
case(SRC_NS + ":115-116", "any", [svt(a3, "double", True)], None, error="does not support")
summarize_cases(a3, "double", SRC_NS + ":117-124", na_bg=True)
# 3-D NaArray with an NA background proper (test-NaArray-matrixStats.R:232-252): col* part
na3 = np.full((6, 5, 4), NA_REAL, dtype=np.float64, order="F")
na3[0, :, 1] = [1e12, -1234.55, -2.1, -1, -0.55]
na3[2, :, 1] = [-0.55, 0, 1e-10, 0.88, 1]
na3[4, :, 1] = [math.pi, 10.33, 3.4567895e8, 300, 2009.01]
na3[5, 2:4, 1] = [0, NAN]
for dims in (1, 2):
    for narm in (False, True):
        kw = {"na_rm": narm, "dims": dims}
        N3 = svt(na3, "double", True)
        case(SRC_NM + ":232-252", "colSums", [N3], stat_col(na3, lambda v: r_sum(v, narm), dims), "equal", kwargs=kw)
        case(SRC_NM + ":232-252", "colMeans", [N3], stat_col(na3, lambda v: r_mean(v, narm), dims), "equal", kwargs=kw)
        case(SRC_NM + ":232-252", "colMins", [N3], stat_col(na3, lambda v: r_minmax_d(v, narm, True), dims), kwargs=kw)
        case(SRC_NM + ":232-252", "colMaxs", [N3], stat_col(na3, lambda v: r_minmax_d(v, narm, False), dims), kwargs=kw)
        case(SRC_NM + ":232-252", "rowSums", [N3], stat_row(na3, lambda v: r_sum(v, narm), dims), "equal", kwargs=kw)
        case(SRC_NM + ":232-252", "rowMins", [N3], stat_row(na3, lambda v: r_minmax_d(v, narm, True), dims), kwargs=kw)
        case(SRC_NM + ":232-252", "rowMaxs", [N3], stat_row(na3, lambda v: r_minmax_d(v, narm, False), dims), kwargs=kw)
# min/max torture, 2-D ints (test-NaArray-matrixStats.R:256-270): col* part
for m in (np.array([[0, -8, NA_INT], [NA_INT, NA_INT, 1]], dtype=np.int32, order="F"),
          np.array([[NA_INT, 0, NA_INT, NA_INT], [8, 9, 1, 1], [-8, -9, -10, -11]], dtype=np.int32, order="F")):
    for narm in (False, True):
        for nm, is_min in (("colMins", True), ("colMaxs", False)):
            case(SRC_NM + ":256-270", nm, [svt(m, "integer", True)],
                 stat_col(m, lambda v: r_minmax_i(v, narm, is_min), dtype=np.int32), kwargs={"na_rm": narm},
                 warn=None)
            case(SRC_NM + ":256-270", "row" + nm[3:], [svt(m, "integer", True)],
                 stat_col(np.asfortranarray(m.T), lambda v: r_minmax_i(v, narm, is_min), dtype=np.int32),
                 kwargs={"na_rm": narm}, warn=None)

# ---------------------------------------------------------------------------
# H2. aperm   tests/testthat/test-SparseArray-aperm.R:42-140 (expected: base::aperm)
# ---------------------------------------------------------------------------
SRC_AP = "tests/testthat/test-SparseArray-aperm.R"
ap2 = np.arange(1, 121, dtype=np.int32).reshape((8, 15), order="F").copy(order="F")
case(SRC_AP + ":44-52", "aperm", [svt(ap2, "integer")], np.asfortranarray(ap2.T))
case(SRC_AP + ":44-52", "aperm", [svt(ap2, "integer")], ap2, kwargs={"perm": [1, 2]})
ap3 = np.arange(1, 361, dtype=np.int32).reshape((8, 3, 15), order="F").copy(order="F")
ap3[:, :, [10, 14]] = 0
ap3[:, 0:2, 13] = 0
ap3[[0, 1, 2, 3, 5], 2, 12] = 0
for perm in ([1, 2, 3], [3, 2, 1], [1, 3, 2], [2, 1, 3], [2, 3, 1], [3, 1, 2]):
    want = np.asfortranarray(np.transpose(ap3, [p - 1 for p in perm]))
    case(SRC_AP + ":54-93", "aperm", [svt(ap3, "integer")], want, kwargs={"perm": perm})
    case(SRC_AP + ":54-93", "aperm", [svt(ap3.astype(np.float64) * 0.5, "double")],
         np.asfortranarray(np.transpose(ap3.astype(np.float64) * 0.5, [p - 1 for p in perm])),
         kwargs={"perm": perm})
ap4 = np.zeros((5, 4, 3, 6), dtype=np.float64, order="F")
ap4 = set_lin(ap4, [1, 17, 18, 60, 61, 119, 200, 201, 202, 300, 359, 360],
              [1.5, -2, NA_REAL, 4, NAN, 6, 7.25, 8, -9, 1e10, 11, 12])
for perm in ([4, 3, 2, 1], [1, 3, 2, 4], [2, 4, 1, 3], [3, 1, 4, 2], [1, 2, 4, 3], [4, 1, 2, 3]):
    case(SRC_AP + ":95-140", "aperm", [svt(ap4, "double")],
         np.asfortranarray(np.transpose(ap4, [p - 1 for p in perm])), kwargs={"perm": perm})
case(SRC_AP + ":95-140", "aperm", [svt(ap4[:, :, 0:0, :], "double")],
     np.asfortranarray(np.transpose(ap4[:, :, 0:0, :], [3, 2, 1, 0])))
# H3. t()   tests/testthat/test-SparseArray-aperm.R:1-38 (expected: base::t; expect_identical).
#     `runif(7, min=-5, max=10)` under set.seed(789) is replaced by seven fixed doubles in that
#     range (R's RNG is unavailable); raw matrices are out of scope.
tm0 = rmat(7, 10)
case(SRC_AP + ":11-13", "t", [svt(tm0, "double")], np.asfortranarray(tm0.T))
tm1 = set_lin(tm0, [5 * k for k in range(1, 15)],
              [-4.21, 9.5, 0.37, 3.3, -1.08, 7.77, 2.5] * 2)
for _c, _v in zip([1, 2, 3, 4, 7, 8, 9], [NA_REAL, NAN, INF, 3e9, 256, -0.999, -1]):
    tm1[1, _c - 1] = _v
case(SRC_AP + ":15-19", "t", [svt(tm1, "double")], np.asfortranarray(tm1.T))
case(SRC_AP + ":21-23", "t", [svt(np.asfortranarray(tm1[0:0, :]), "double")],
     np.asfortranarray(tm1[0:0, :].T))
case(SRC_AP + ":24-26", "t", [svt(np.asfortranarray(tm1[:, 0:0]), "double")],
     np.asfortranarray(tm1[:, 0:0].T))
with np.errstate(all="ignore"):
    tm1i = np.where(np.isfinite(tm1) & (np.abs(tm1) < 2 ** 31), np.trunc(np.nan_to_num(tm1, posinf=0, neginf=0)), NA_INT).astype(np.int32)
tm1i[~np.isfinite(tm1) | (np.abs(tm1) >= 2 ** 31)] = NA_INT      # storage.mode(m) <- "integer": NA with a warning
case(SRC_AP + ":29-31", "t", [svt(tm1i, "integer")], np.asfortranarray(tm1i.T))
tm1l = np.where(np.isnan(tm1), NA_INT, (tm1 != 0).astype(np.int32)).astype(np.int32)
case(SRC_AP + ":32-34", "t", [svt(tm1l, "logical")], np.asfortranarray(tm1l.T))
# row statistics of >2-D arrays that have no native kernel go through aperm
# (R/SparseArray-matrixStats.R:115-118)
for dims in (1, 2):
    case(SRC_MS + ":244-270", "rowProds", [svt(a3_clean, "double")],
         stat_row(a3_clean, lambda v: r_prod(v, False), dims), "equal", kwargs={"dims": dims})
    case(SRC_MS + ":244-270", "rowMeans", [svt(a3, "double")],
         stat_row(a3, lambda v: r_mean(v, True), dims), "equal", kwargs={"dims": dims, "na_rm": True})

# ---------------------------------------------------------------------------
# I. rowsum / colsum   tests/testthat/test-rowsum-methods.R:61-89
# ---------------------------------------------------------------------------
SRC_RS = "tests/testthat/test-rowsum-methods.R"


def rowsum_dense(m, group, reorder, narm):
    ug = list(dict.fromkeys(group))
    if reorder:
        ug = sorted(ug)
    isint = m.dtype == np.int32
    out = np.zeros((len(ug), m.shape[1]), dtype=m.dtype, order="F")
    for gi, g in enumerate(ug):
        rows = [i for i, x in enumerate(group) if x == g]
        for j in range(m.shape[1]):
            v = m[rows, j]
            if isint:
                vv = vals_i(v, narm)
                out[gi, j] = NA_INT if has_na(vv) else int(vv.sum())
            else:
                out[gi, j] = r_sum(v, narm)
    return out


def to_dgc(m):
    p = [0]
    ii, xx = [], []
    for j in range(m.shape[1]):
        nz = np.flatnonzero(m[:, j] != 0)
        ii += list(nz)
        xx += list(m[nz, j])
        p.append(len(ii))
    return {"__dgc__": True, "dim": list(m.shape), "p": enc(np.array(p, np.int32)),
            "i": enc(np.array(ii, np.int32)), "x": enc(np.array(xx, np.float64))}


def rowsum_cases(m, group, src):
    t = "double" if m.dtype == np.float64 else "integer"
    cmp = "equal" if t == "double" else "identical"
    tm = np.asfortranarray(m.T)
    for reorder in (True, False):
        for narm in (False, True):
            kw = {"group": list(group), "reorder": reorder, "na_rm": narm}
            exp = rowsum_dense(m, group, reorder, narm)
            case(src, "rowsum", [svt(m, t)], exp, cmp, kwargs=kw)
            case(src, "colsum", [svt(tm, t)], np.asfortranarray(exp.T), cmp, kwargs=kw)
            md = m if t == "double" else int_to_double(m)
            expd = exp if t == "double" else int_to_double(exp)
            case(src, "rowsum", [to_dgc(md)], expd, "equal", kwargs=kw)
            case(src, "colsum", [to_dgc(np.asfortranarray(md.T))], np.asfortranarray(expd.T), "equal", kwargs=kw)


grp = ["B", "A", "B", "B", "B", "A"]
rs0 = rmat(6, 4)
rowsum_cases(rs0, grp, SRC_RS + ":63-67")
rs1 = rs0.copy()
rs1[:, 0] = [8.55, INF, NA_REAL, 0, NAN, -INF]
rs1[:, 2] = [0.6, -11.99, 0, 4.44, 0, 0]
rs1[:, 3] = [1, 2, 3, 4, 5, 6]
rowsum_cases(rs1, grp, SRC_RS + ":69-73")
rowsum_cases(np.asfortranarray(rs1[0:0, :]), [], SRC_RS + ":74-75")
rowsum_cases(np.asfortranarray(rs1[:, 0:0]), grp, SRC_RS + ":76-77")
rowsum_cases(np.asfortranarray(rs1[0:0, 0:0]), [], SRC_RS + ":78-79")
rs2 = rmat(6, 4, 0, np.int32)
rs2[0, 1] = NA_INT
rs2[2, 1] = 99
rs2[:, 3] = [1, 2, 3, 4, 5, 6]
rowsum_cases(rs2, grp, SRC_RS + ":81-88")

# ---------------------------------------------------------------------------
# J. man/SparseArray-matrixStats.Rd:189-227
# ---------------------------------------------------------------------------
SRC_MAN = "man/SparseArray-matrixStats.Rd:189-227"
d0 = set_lin(rmat(6, 4, 0, np.int32), [1, 2, 8, 10, 15, 16, 17, 24], [k * 10 for k in range(1, 9)])
d0[4, 1] = NA_INT
D0 = svt(d0, "integer")
td0 = np.asfortranarray(d0.T)
for narm in (False, True):
    kw = {"na_rm": narm}
    case(SRC_MAN, "colSums", [D0], stat_col(d0, lambda v: r_sum(v, narm)), kwargs=kw)
    case(SRC_MAN, "rowSums", [D0], stat_col(td0, lambda v: r_sum(v, narm)), kwargs=kw)
    case(SRC_MAN, "colMeans", [D0], stat_col(d0, lambda v: r_mean(v, narm)), kwargs=kw)
    lo = stat_col(d0, lambda v: r_minmax_i(v, narm, True), dtype=np.int32)
    hi = stat_col(d0, lambda v: r_minmax_i(v, narm, False), dtype=np.int32)
    case(SRC_MAN, "colRanges", [D0], np.stack([lo, hi], axis=-1), kwargs=kw)
    case(SRC_MAN, "colVars", [D0], stat_col(d0, lambda v: r_var(v, narm)), "equal", kwargs=kw)


# ---------------------------------------------------------------------------
# K. tests/testthat/test-sparseMatrix-utils.R:23-88  colStats_dgCMatrix
#    The test draws `rsparsematrix(22, 10, density=0.25)` under set.seed(123); R's RNG is
#    not available, so a fixed 22 x 10 matrix of the same density stands in (55 nonzeros at
#    positions and values written out below); the NA / NaN edits of :57-64 are the test's own.
#    Expected: apply(m, 2, min) etc. on the dense matrix (:29-55); colMins/colMaxs/colRanges
#    compared with expect_identical, colVars with expect_equal.
# ---------------------------------------------------------------------------
SRC_DGC = "tests/testthat/test-sparseMatrix-utils.R"
_dg = rmat(22, 10)
_pos = [(3 * k * k + 7 * k + 1) % 220 for k in range(1, 80)]
_seen = []
for _q in _pos:
    if _q not in _seen:
        _seen.append(_q)
    if len(_seen) == 55:
        break
for _n, _q in enumerate(_seen):
    _dg[_q % 22, _q // 22] = round(((_n * 37) % 41 - 20) / 8.0 + (0.13 if _n % 3 == 0 else -0.07), 2)
_dg[_dg == 0] = 0.0
_dg[2, 0] = NA_REAL      # m0[3L, 1L] <- NA
_dg[2, 1] = NAN          # m0[3L, 2L] <- NaN
_dg[3, 3] = NA_REAL      # m0[4L, 4L] <- NA
_dg[7, 3] = NAN          # m0[8L, 4L] <- NaN
_dg[3, 7] = NAN          # m0[4L, 8L] <- NaN
_dg[21, 7] = NA_REAL     # m0[22L, 8L] <- NA
_dg[:, 8] = NA_REAL      # m0[ , 9L] <- NA
_dg[:, 9] = NAN          # m0[ , 10L] <- NaN


def dgc_colstat_cases(m, src):
    g = to_dgc(m)
    for narm in (False, True):
        kw = {"na_rm": narm}
        lo = col_apply(m, lambda v: r_minmax_d(v, narm, True))
        hi = col_apply(m, lambda v: r_minmax_d(v, narm, False))
        case(src + ":29-34", "colMins_dgCMatrix", [g], lo, kwargs=kw)
        case(src + ":36-42", "colMaxs_dgCMatrix", [g], hi, kwargs=kw)
        case(src + ":44-50", "colRanges_dgCMatrix", [g], np.stack([lo, hi], axis=-1), kwargs=kw)
        case(src + ":52-58", "colVars_dgCMatrix", [g], col_apply(m, lambda v: r_var(v, narm)),
             "equal", kwargs=kw)


dgc_colstat_cases(_dg, SRC_DGC)
# the same statistics on the rowsum fixtures of test-rowsum-methods.R:69-73 (Inf, -Inf, NA, NaN
# in one column; a column with no stored entry)
dgc_colstat_cases(rs1, "tests/testthat/test-rowsum-methods.R:69-73 + " + SRC_DGC)


def main():
    out = os.path.join(HERE, "golden.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "ncases": len(CASES), "cases": CASES}, f, separators=(",", ":"))
    print(f"wrote {out}: {len(CASES)} cases, {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    main()
