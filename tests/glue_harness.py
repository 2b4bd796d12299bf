"""TEST INFRASTRUCTURE ONLY (build container, CPU): integration/svt_hip_glue.c compiled, linked and RUN.

The glue is the one product artifact that needs R to execute; the image has none.  This module builds it
into a shared library together with
  * tests/r_api_standin/r_standin.c -- a functional test-only stand-in for the part of R's C API it uses
    (fake SEXPs, attributes, a COUNTED protection stack, guarded R_alloc() blocks, error() as a longjmp,
    warning() into a log),
  * the reference helper files the glue calls that compile as they are, straight from the read-only mount
    (src/argcheck_utils.c, src/Rvector_utils.c, src/Rvector_summarization.c -- hidden visibility +
    --gc-sections: only what the glue reaches is kept),
  * tests/r_api_standin/glue_env.c -- the helpers that are `static` in the reference (its maintainer makes
    them extern for the glue), the leaf constructor, and recording stubs for the `_cpu` bodies,
and binds the svt_* symbols the glue dlsym()s to the CPU oracle's ABI (tests/r_api_standin/svt_over_oracle.c).
`GlueDispatcher` then offers the `.Call`-level interface of sparsearray_amd/_dispatch.py THROUGH the
registered C_* names, so that `Session(GlueDispatcher(...))` pushes the golden cases through the glue.
Nothing here travels to the GPU box as a product path; /root/reference must be present.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import warnings

import numpy as np

from sparsearray_amd.api import OPCODES, SparseArrayError
from sparsearray_amd.svt import SVT_SparseArray

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
STANDIN = os.path.join(ROOT, "tests", "r_api_standin")
REF_HELPERS = ("argcheck_utils", "Rvector_utils", "Rvector_summarization")

LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP = 10, 13, 14, 16, 19
_NP = {LGLSXP: np.int32, INTSXP: np.int32, REALSXP: np.float64}


def build(outdir: str) -> tuple[str, str]:
    """Returns (harness .so, shim .so).  Raises CalledProcessError with the compiler's text on failure."""
    oracle_dir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-s", "-C", oracle_dir])
    inc = ["-I", STANDIN, "-I", REF_SRC, "-I", os.path.join(ROOT, "include"), "-I", oracle_dir]
    objs = []

    def cc(src, extra=()):
        obj = os.path.join(outdir, os.path.basename(src)[:-2] + ".o")
        cmd = ["gcc", "-c", "-O1", "-g", "-fPIC", *extra, *inc, src, "-o", obj]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{res.stderr}")
        objs.append(obj)
    cc(os.path.join(ROOT, "integration", "svt_hip_glue.c"), ["-Wall", "-Werror=implicit-function-declaration"])
    cc(os.path.join(STANDIN, "r_standin.c"), ["-Wall"])
    cc(os.path.join(STANDIN, "glue_env.c"), ["-Wall"])
    for h in REF_HELPERS:       # the reference's files, read where they lie; nothing of them is kept in the repository
        cc(os.path.join(REF_SRC, h + ".c"), ["-fvisibility=hidden", "-ffunction-sections", "-fdata-sections", "-w"])
    harness = os.path.join(outdir, "libglue_harness.so")
    shim = os.path.join(outdir, "libsvt_shim.so")
    link = ["gcc", "-shared", "-o", harness, *objs, "-Wl,--gc-sections", "-Wl,-z,defs", "-L", oracle_dir, "-lsvt_oracle",
            f"-Wl,-rpath,{oracle_dir}", "-ldl", "-lm"]
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"{' '.join(link)}\n{res.stderr}")
    cmd = ["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Werror", *inc, os.path.join(STANDIN, "svt_over_oracle.c"), "-o", shim,
           "-L", oracle_dir, "-lsvt_oracle", f"-Wl,-rpath,{oracle_dir}"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)}\n{res.stderr}")
    return harness, shim


class RError(SparseArrayError):
    """error() raised inside a .Call."""


class Glue:
    """The harness library + constructors / readers of fake SEXPs."""

    def __init__(self, harness: str, shim: str):
        os.environ["SPARSEARRAY_HIP_LIB"] = shim            # what the glue dlopen()s instead of libsvt_hip.so
        self.lib = L = ctypes.CDLL(harness)
        P = ctypes.c_void_p
        for name, res, args in (
            ("sx_nil", P, []), ("sx_alloc", P, [ctypes.c_int, ctypes.c_long]), ("sx_data", P, [P]),
            ("sx_len", ctypes.c_long, [P]), ("sx_type", ctypes.c_int, [P]), ("sx_symbol", P, [ctypes.c_char_p]),
            ("sx_mkchar", P, [ctypes.c_char_p]), ("sx_na_string", P, []), ("sx_char", ctypes.c_char_p, [P]),
            ("sx_nattr", ctypes.c_int, [P]), ("sx_attr_name", ctypes.c_char_p, [P, ctypes.c_int]),
            ("sx_attr_value", P, [P, ctypes.c_int]), ("sx_call", ctypes.c_int, [P, ctypes.c_int, P, P]),
            ("sx_error", ctypes.c_char_p, []), ("sx_warnings", ctypes.c_char_p, []), ("sx_nwarnings", ctypes.c_int, []),
            ("sx_protect_depth", ctypes.c_int, []), ("sx_protect_max", ctypes.c_int, []),
            ("sx_protect_underflow", ctypes.c_int, []), ("sx_guards_broken", ctypes.c_int, []), ("sx_reset", None, []),
            ("Rf_setAttrib", P, [P, P, P]), ("SET_VECTOR_ELT", P, [P, ctypes.c_long, P]),
            ("SET_STRING_ELT", None, [P, ctypes.c_long, P]), ("VECTOR_ELT", P, [P, ctypes.c_long]),
            ("STRING_ELT", P, [P, ctypes.c_long]),
            ("env_cpu_body_calls", ctypes.c_int, []), ("env_cpu_body_last", ctypes.c_char_p, []),
            ("env_cpu_body_reset", None, []),
        ):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        self.nil = L.sx_nil()
        self.stats = {"calls": 0, "max_protect": 0, "errors": 0, "warnings": 0}

    # ---- Python -> SEXP ---------------------------------------------------------------------
    def vec(self, a, sxtype=None):
        a = np.asarray(a)
        if sxtype is None:
            sxtype = REALSXP if a.dtype == np.float64 else LGLSXP if a.dtype == np.bool_ else INTSXP
        flat = np.ascontiguousarray(np.reshape(a, -1, order="F"), dtype=_NP[sxtype])
        s = self.lib.sx_alloc(sxtype, flat.size)
        if flat.size:
            ctypes.memmove(self.lib.sx_data(s), flat.ctypes.data, flat.nbytes)
        return s

    def ints(self, *v):
        return self.vec(np.asarray(v, dtype=np.int32), INTSXP)

    def lgl(self, b):
        return self.vec(np.asarray([int(bool(b))], dtype=np.int32), LGLSXP)

    def real(self, x):
        return self.vec(np.asarray([x], dtype=np.float64), REALSXP)

    def string(self, *strs):
        s = self.lib.sx_alloc(STRSXP, len(strs))
        for i, t in enumerate(strs):
            self.lib.SET_STRING_ELT(s, i, self.lib.sx_na_string() if t is None else self.lib.sx_mkchar(str(t).encode()))
        return s

    def rlist(self, items):
        s = self.lib.sx_alloc(VECSXP, len(items))
        for i, it in enumerate(items):
            self.lib.SET_VECTOR_ELT(s, i, it)
        return s

    def set_attr(self, s, name, v):
        self.lib.Rf_setAttrib(s, self.lib.sx_symbol(name.encode()), v)
        return s

    def matrix(self, a):
        a = np.asarray(a)
        s = self.vec(a)
        return self.set_attr(s, "dim", self.ints(*a.shape))

    def dimnames(self, dn):
        if dn is None:
            return self.nil
        return self.rlist([self.nil if d is None else self.string(*d) for d in dn])

    def svt_tree(self, x: SVT_SparseArray):
        """x@SVT: nested lists over dims N..2, leaves list(nzvals | NULL, nzoffs); subtrees with no
        nonzero are NULL (what the reference's constructors produce)."""
        if x.svt_is_null:
            return self.nil

        def leaf(lf):
            if lf is None:
                return self.nil
            offs, vals = lf
            sx = INTSXP if x.type == "integer" else LGLSXP if x.type == "logical" else REALSXP
            return self.rlist([self.nil if vals is None else self.vec(np.asarray(vals), sx),
                               self.vec(np.asarray(offs, dtype=np.int32), INTSXP)])

        def rec(first, ndim):
            if ndim == 1:
                return leaf(x.leaves[first])
            stride = int(np.prod(x.dim[1:ndim - 1], dtype=np.int64))
            n = x.dim[ndim - 1]
            if all(lf is None for lf in x.leaves[first:first + stride * n]):
                return self.nil
            return self.rlist([rec(first + i * stride, ndim - 1) for i in range(n)])
        return rec(0, x.ndim)

    def svt_args(self, x: SVT_SparseArray):
        return self.ints(*x.dim), self.string(x.type), self.svt_tree(x)

    def dgc(self, x):
        (nrow, ncol), p, i, xx = x
        s = self.lib.sx_alloc(VECSXP, 0)
        self.set_attr(s, "Dim", self.ints(nrow, ncol))
        self.set_attr(s, "p", self.vec(np.asarray(p, dtype=np.int32), INTSXP))
        self.set_attr(s, "i", self.vec(np.asarray(i, dtype=np.int32), INTSXP))
        self.set_attr(s, "x", self.vec(np.asarray(xx, dtype=np.float64), REALSXP))
        return s

    # ---- SEXP -> Python ---------------------------------------------------------------------
    def attrs(self, s) -> dict:
        return {self.lib.sx_attr_name(s, i).decode(): self.lib.sx_attr_value(s, i) for i in range(self.lib.sx_nattr(s))}

    def to_numpy(self, s):
        t, n = self.lib.sx_type(s), self.lib.sx_len(s)
        a = np.ctypeslib.as_array(ctypes.cast(self.lib.sx_data(s), ctypes.POINTER(
            ctypes.c_double if t == REALSXP else ctypes.c_int32)), shape=(max(n, 1),))[:n].copy()
        at = self.attrs(s)
        if "dim" in at:
            a = a.reshape(tuple(self.to_numpy(at["dim"]).tolist()), order="F")
        return a

    def strings(self, s):
        if s == self.nil or s is None:
            return None
        return [self.lib.sx_char(self.lib.STRING_ELT(s, i)).decode() for i in range(self.lib.sx_len(s))]

    def names_of(self, s):
        at = self.attrs(s)
        if "dimnames" in at:
            dn = at["dimnames"]
            return [self.strings(self.lib.VECTOR_ELT(dn, i)) for i in range(self.lib.sx_len(dn))]
        if "names" in at:
            return [self.strings(at["names"])]
        return None

    def svt_from_tree(self, tree, dim, type_) -> SVT_SparseArray:
        """The R tree an entry point returned -> SVT_SparseArray (checks the shape of every node on the way)."""
        nleaves = int(np.prod(dim[1:], dtype=np.int64)) if len(dim) > 1 else 1
        leaves = [None] * nleaves
        npd = np.float64 if type_ == "double" else np.int32

        def rec(node, first, ndim):
            if node == self.nil or node is None:
                return
            assert self.lib.sx_type(node) == VECSXP
            if ndim == 1:
                assert self.lib.sx_len(node) == 2, "a leaf is list(nzvals, nzoffs)"
                vals, offs = self.lib.VECTOR_ELT(node, 0), self.lib.VECTOR_ELT(node, 1)
                o = self.to_numpy(offs).astype(np.int32)
                assert self.lib.sx_type(offs) == INTSXP and o.size > 0
                v = None if vals == self.nil else self.to_numpy(vals).astype(npd)
                assert v is None or v.size == o.size
                leaves[first] = (o, v)
                return
            stride = int(np.prod(dim[1:ndim - 1], dtype=np.int64))
            assert self.lib.sx_len(node) == dim[ndim - 1]
            for i in range(dim[ndim - 1]):
                rec(self.lib.VECTOR_ELT(node, i), first + i * stride, ndim - 1)
        rec(tree, 0, len(dim))
        return SVT_SparseArray(dim, type_, leaves)

    # ---- one .Call ---------------------------------------------------------------------------
    def call(self, name: str, *args, allow_cpu_body: bool = False):
        """Returns the result SEXP; error() -> RError; warning()s are re-issued as Python warnings.  Every call is
        checked for R's protection discipline (depth 0 at exit, never popped below 0) and for R_alloc() overruns."""
        f = ctypes.cast(getattr(self.lib, name), ctypes.c_void_p)
        arr = (ctypes.c_void_p * max(len(args), 1))(*args)
        out = ctypes.c_void_p(0)
        self.lib.env_cpu_body_reset()
        failed = self.lib.sx_call(f, len(args), arr, ctypes.byref(out))
        self.stats["calls"] += 1
        self.stats["max_protect"] = max(self.stats["max_protect"], self.lib.sx_protect_max())
        assert self.lib.sx_guards_broken() == 0, f"{name}: wrote past an R_alloc() block"
        assert self.lib.sx_protect_underflow() == 0, f"{name}: UNPROTECT() below the depth at entry"
        if failed:
            self.stats["errors"] += 1
            raise RError(self.lib.sx_error().decode())
        assert self.lib.sx_protect_depth() == 0, f"{name}: returned with {self.lib.sx_protect_depth()} object(s) protected"
        assert allow_cpu_body or self.lib.env_cpu_body_calls() == 0 or name.endswith("_threads") or name.endswith("_procs"), \
            f"{name}: fell through to {self.lib.env_cpu_body_last().decode()} with the library available"
        nw = self.lib.sx_nwarnings()
        msgs = self.lib.sx_warnings().decode().split("\x1e") if nw else []
        self.stats["warnings"] += nw
        self.last_warnings = msgs
        return out.value

    def reset(self):
        self.lib.sx_reset()


class GlueDispatcher:
    """The `.Call` dispatcher of sparsearray_amd/api.py's Session, through the glue's registered C_* names."""
    accepts_mixed_types = False

    def __init__(self, glue: Glue, fallback):
        self.g = glue
        self.fallback = fallback             # the oracle's dispatcher, for entry points the glue does not register

    def __call__(self, name, *args):
        if not hasattr(self, name):
            return self.fallback(name, *args)
        try:
            return getattr(self, name)(*args)
        finally:
            self.g.reset()

    def has_entry(self, name):
        return hasattr(self.g.lib, name)

    def _opcode(self, op):
        if op not in OPCODES:
            raise SparseArrayError("'op' must be one of: " + ", ".join(f'"{k}"' for k in OPCODES))
        return OPCODES[op]

    def _warn(self):
        return any("NAs introduced" in m or "integer overflow" in m for m in self.g.last_warnings)

    # thread control
    def C_get_num_procs(self):
        return int(self.g.to_numpy(self.g.call("C_get_num_procs"))[0])

    def C_get_max_threads(self):
        return int(self.g.to_numpy(self.g.call("C_get_max_threads"))[0])

    def C_set_max_threads(self, n):
        return int(self.g.to_numpy(self.g.call("C_set_max_threads", self.g.ints(int(n))))[0])

    # crossprod (src/SparseMatrix_mult.h:6-43)
    def C_crossprod2_SVT_mat(self, x, y, tr_y):
        g = self.g
        y = np.asarray(y)
        ans = g.call("C_crossprod2_SVT_mat", *g.svt_args(x), g.matrix(y), g.lgl(tr_y), g.string("double"),
                     g.dimnames([["r%d" % i for i in range(x.dim[1])], None]))
        out = g.to_numpy(ans)
        assert g.names_of(ans) == [["r%d" % i for i in range(x.dim[1])], None]      # ans_dimnames attached as given
        return out

    def C_crossprod2_mat_SVT(self, x, y, tr_x):
        g = self.g
        x = np.asarray(x)
        ans = g.call("C_crossprod2_mat_SVT", g.matrix(x), *g.svt_args(y), g.lgl(tr_x), g.string("double"), g.nil)
        return g.to_numpy(ans)

    def C_crossprod2_SVT_SVT(self, x, y):
        g = self.g
        ans = g.call("C_crossprod2_SVT_SVT", *g.svt_args(x), *g.svt_args(y), g.string("double"), g.nil)
        return g.to_numpy(ans)

    def C_crossprod1_SVT(self, x):
        g = self.g
        return g.to_numpy(g.call("C_crossprod1_SVT", *g.svt_args(x), g.string("double"), g.nil))

    # statistics (src/SparseArray_matrixStats.h:6-28, src/SparseArray_summarization.h)
    def C_summarize_SVT(self, x, op, na_rm, center):
        g = self.g
        self._opcode(op)
        xd, xt, xs = g.svt_args(x)
        ans = g.call("C_summarize_SVT", xd, xt, xs, g.lgl(x.na_background), g.string(op), g.lgl(na_rm), g.real(center))
        a = g.to_numpy(ans)
        t = g.lib.sx_type(ans)
        val = a if op == "range" else (np.float64(a[0]) if t == REALSXP else np.int32(a[0]))
        return val, self._warn()

    def _stats(self, name, x, op, na_rm, center_sexp, dims, shape, names_want):
        g = self.g
        self._opcode(op)
        xd, xt, xs = g.svt_args(x)
        ans = g.call(name, xd, g.dimnames(x.dimnames), xt, xs, g.lgl(x.na_background), g.string(op), g.lgl(na_rm),
                     center_sexp, g.ints(int(dims)))
        a = g.to_numpy(ans)
        assert a.shape == (tuple(shape) if len(shape) > 1 else (int(np.prod(shape, dtype=np.int64)) if shape else 1,)), \
            (name, a.shape, shape)
        if x.dimnames is not None:                       # names / dimnames propagated as the reference does
            kept = list(names_want)
            got = g.names_of(ans)
            if len(kept) == 0 or all(k is None for k in kept):
                assert got is None, (name, got)
            else:
                assert got == [None if k is None else [str(t) for t in k] for k in kept], (name, got, kept)
        return a, self._warn()

    def C_colStats_SVT(self, x, op, na_rm, center, dims):
        shape = tuple(x.dim[dims:])
        want = [] if x.dimnames is None else list(x.dimnames[dims:])
        return self._stats("C_colStats_SVT", x, op, na_rm, self.g.real(center), dims, shape, want)

    def C_rowStats_SVT(self, x, op, na_rm, center, dims):
        g = self.g
        shape = tuple(x.dim[:dims])
        csexp = g.nil
        if center is not None:
            c = np.reshape(np.asarray(center, np.float64), -1, order="F")
            csexp = g.vec(c, REALSXP)
        want = [] if x.dimnames is None else list(x.dimnames[:dims])
        return self._stats("C_rowStats_SVT", x, op, na_rm, csexp, dims, shape, want)

    # rowsum / colsum (src/rowsum_methods.h:6-36)
    def _groupsum(self, name, x, group, ngroup, na_rm):
        g = self.g
        ans = g.call(name, *g.svt_args(x), g.vec(np.asarray(group, dtype=np.int32), INTSXP), g.ints(int(ngroup)), g.lgl(na_rm))
        return g.to_numpy(ans), self._warn()

    def C_rowsum_SVT(self, x, group, ngroup, na_rm):
        return self._groupsum("C_rowsum_SVT", x, group, ngroup, na_rm)

    def C_colsum_SVT(self, x, group, ngroup, na_rm):
        return self._groupsum("C_colsum_SVT", x, group, ngroup, na_rm)

    def _groupsum_dgc(self, name, x, group, ngroup, na_rm):
        g = self.g
        ans = g.call(name, g.dgc(x), g.vec(np.asarray(group, dtype=np.int32), INTSXP), g.ints(int(ngroup)), g.lgl(na_rm))
        return g.to_numpy(ans)

    def C_rowsum_dgCMatrix(self, x, group, ngroup, na_rm):
        return self._groupsum_dgc("C_rowsum_dgCMatrix", x, group, ngroup, na_rm)

    def C_colsum_dgCMatrix(self, x, group, ngroup, na_rm):
        return self._groupsum_dgc("C_colsum_dgCMatrix", x, group, ngroup, na_rm)

    def _dgc_stat(self, name, x, na_rm):
        g = self.g
        return g.to_numpy(g.call(name, g.dgc(x), g.lgl(na_rm)))

    def C_colMins_dgCMatrix(self, x, na_rm):
        return self._dgc_stat("C_colMins_dgCMatrix", x, na_rm)

    def C_colMaxs_dgCMatrix(self, x, na_rm):
        return self._dgc_stat("C_colMaxs_dgCMatrix", x, na_rm)

    def C_colRanges_dgCMatrix(self, x, na_rm):
        return self._dgc_stat("C_colRanges_dgCMatrix", x, na_rm)

    def C_colVars_dgCMatrix(self, x, na_rm):
        return self._dgc_stat("C_colVars_dgCMatrix", x, na_rm)

    # t() / aperm() (src/SparseArray_aperm.h)
    def C_transpose_2D_SVT(self, x):
        g = self.g
        tree = g.call("C_transpose_2D_SVT", *g.svt_args(x))
        ans = g.svt_from_tree(tree, (x.dim[1], x.dim[0]), x.type)
        if x.dimnames is not None:
            ans.dimnames = [x.dimnames[1], x.dimnames[0]]
        ans.na_background = x.na_background
        return ans

    def C_aperm_SVT(self, x, perm):
        g = self.g
        perm = [int(p) for p in np.asarray(perm).reshape(-1)]
        tree = g.call("C_aperm_SVT", *g.svt_args(x), g.ints(*perm))
        new_dim = tuple(x.dim[p - 1] for p in perm)
        ans = g.svt_from_tree(tree, new_dim, x.type)
        if x.dimnames is not None:
            ans.dimnames = [x.dimnames[p - 1] for p in perm]
        ans.na_background = x.na_background
        return ans


def reissue(msgs):
    for m in msgs:
        warnings.warn(m)
