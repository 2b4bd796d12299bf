"""integration/svt_hip_glue.c meets a compiler: syntax, types and arities of
the R-facing glue are checked with gcc -fsyntax-only against the REFERENCE's own headers (read where they
lie, /root/reference/src -- absent on the GPU box, so the test skips there) and a declarations-only,
test-only stand-in for R's <Rdefines.h> (tests/r_api_standin/; no definitions, never shipped or linked).
Interface checked: src/R_init_SparseArray.c:31-147, src/SparseMatrix_mult.h:6-43,
src/SparseArray_matrixStats.h:6-28, src/rowsum_methods.h:6-36."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
GLUE = os.path.join(ROOT, "integration", "svt_hip_glue.c")


@pytest.mark.skipif(not os.path.isdir(REF_SRC) or shutil.which("gcc") is None,
                    reason="needs the reference's headers and gcc (build container only)")
def test_glue_passes_the_compiler():
    cmd = ["gcc", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration",
           "-Werror=int-conversion", "-Werror=incompatible-pointer-types", "-Werror=return-type",
           "-I", os.path.join(ROOT, "tests", "r_api_standin"), "-I", REF_SRC,
           "-I", os.path.join(ROOT, "include"), GLUE]
    res = subprocess.run(cmd, capture_output=True, text=True)
    own = [ln for ln in res.stderr.splitlines() if "svt_hip_glue.c:" in ln and ("warning" in ln or "error" in ln)]
    assert res.returncode == 0 and not own, res.stderr


def _functions(src):
    """(name, nargs, body) of every function DEFINED in the glue."""
    out = []
    for m in re.finditer(r"^(?:static\s+)?(?:SEXP|int|void|const\s+\w+\s*\*|\w+)\s+\**(\w+)\(([^;{]*?)\)\s*\{", src, re.M):
        depth, i = 1, m.end()
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        args = m.group(2).strip()
        nargs = 0 if args in ("", "void") else args.count(",") + 1
        out.append((m.group(1), nargs, src[m.end():i]))
    return out


def test_glue_entry_points_have_the_registered_arity():
    """Names and argument counts of src/R_init_SparseArray.c:41-43,49-52,70,72,94,121-134."""
    want = {"C_crossprod2_SVT_mat": 7, "C_crossprod2_mat_SVT": 7, "C_crossprod2_SVT_SVT": 8, "C_crossprod1_SVT": 5,
            "C_colStats_SVT": 9, "C_rowStats_SVT": 9, "C_summarize_SVT": 7, "C_rowsum_SVT": 6, "C_colsum_SVT": 6,
            "C_rowsum_dgCMatrix": 4, "C_colsum_dgCMatrix": 4, "C_get_num_procs": 0, "C_get_max_threads": 0,
            "C_set_max_threads": 1, "C_transpose_2D_SVT": 3, "C_aperm_SVT": 4, "C_colMins_dgCMatrix": 2,
            "C_colMaxs_dgCMatrix": 2, "C_colRanges_dgCMatrix": 2, "C_colVars_dgCMatrix": 2}
    got = {n: a for n, a, _ in _functions(open(GLUE).read()) if n.startswith("C_")}
    assert got == want


def _check_protect_balance(name, body):
    """Walks a function body keeping R's protection depth: PROTECT pushes, UNPROTECT(n) pops n (never more
    than the function pushed), every `return` must see depth 0, a block that returns leaves the depth of the
    code after it untouched, and the alternatives of an if / else chain must end at the same depth."""
    depth = 0
    frames = []                     # per open block: depth at entry, "has returned", end depths of earlier alternatives
    alts_next = None
    for m in re.finditer(r"\bPROTECT\(|\bUNPROTECT\((\d+)\)|\{|\}|\breturn\b", body):
        t = m.group(0)
        if t.startswith("PROTECT"):
            depth += 1
        elif t.startswith("UNPROTECT"):
            depth -= int(m.group(1))
            assert depth >= 0, f"{name}: UNPROTECT({m.group(1)}) pops more than was protected"
        elif t == "return":
            assert depth == 0, f"{name}: return with {depth} object(s) still protected"
            if frames:
                frames[-1]["returned"] = True
        elif t == "{":
            frames.append({"entry": depth, "returned": False, "alts": alts_next or []})
            alts_next = None
        elif t == "}":
            if not frames:
                break                                   # the function's own closing brace
            blk = frames.pop()
            ends = blk["alts"] + ([] if blk["returned"] else [depth])
            if body[m.end():].lstrip().startswith("else"):
                alts_next = ends                        # the next alternative starts where this one did
                depth = blk["entry"]
            else:
                assert len(set(ends)) <= 1, f"{name}: branches end at different protection depths {ends}"
                depth = ends[0] if ends else blk["entry"]
                if blk["returned"] and not blk["alts"]:
                    depth = blk["entry"]
    assert depth == 0, f"{name}: falls off its end with {depth} object(s) protected"


def test_glue_protect_balance():
    """PROTECT / UNPROTECT discipline of every function of the glue (R's protection stack must be
    balanced on every path out of a .Call entry point)."""
    for name, _, body in _functions(open(GLUE).read()):
        _check_protect_balance(name, body)
