"""Wall time of host-level entry points of the C ABI on SMALL operands (what an R session with many small calls sees;
the view is built once, as the R glue's make_view() costs microseconds): BASELINE config 1 (1e4 x 1e3 @ 1 %) colSums /
colVars / rowSums, and the reference's published crossprod shape, libsvt_hip.so vs the CPU oracle (same ABI).
    python tools/debug/host_call_latency.py [reps]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
from sparsearray_amd import _hip
from sparsearray_amd.svt import make_view_from_csc
from helpers import random_csc
from oracle import load_oracle
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
hl, orc = _hip.init(), load_oracle()
P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
for lib, pre in ((hl, "svt_"), (orc, "orc_")):
    getattr(lib, pre + "colStats_SVT").argtypes = [P, I, I, D, I, P, P]
    getattr(lib, pre + "rowStats_SVT").argtypes = [P, I, I, P, I, P, P]
    getattr(lib, pre + "crossprod1_SVT").argtypes = [P, P]
    getattr(lib, pre + "crossprod2_SVT_mat").argtypes = [P, P, I, I, I, I, P]


def best(fn):
    fn(); fn()
    b = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); rc = fn(); b = min(b, time.perf_counter() - t0)
        assert rc == 0
    return b * 1e3


def table(tag):
    cp, ri, v = random_csc(10_000, 1_000, 0.01, seed=1)
    view = make_view_from_csc((10_000, 1_000), "double", cp, ri, v)
    out = np.zeros(10_000); warn = ctypes.c_int(0)
    y = np.ascontiguousarray(np.random.default_rng(2).uniform(-1, 1, (16, 10_000))); o2 = np.zeros((16, 1_000))
    cp1, ri1, v1 = random_csc(25_000, 400, 0.07, seed=11)
    v1w = make_view_from_csc((25_000, 400), "double", cp1, ri1, v1); o3 = np.zeros((400, 400))
    rows = [("colSums (op 8)", lambda L, p: getattr(L, p + "colStats_SVT")(ctypes.addressof(view), 8, 0, float("nan"), 1, out.ctypes.data, ctypes.byref(warn))),
            ("colVars (op 13)", lambda L, p: getattr(L, p + "colStats_SVT")(ctypes.addressof(view), 13, 0, float("nan"), 1, out.ctypes.data, ctypes.byref(warn))),
            ("rowSums (op 8)", lambda L, p: getattr(L, p + "rowStats_SVT")(ctypes.addressof(view), 8, 0, None, 1, out.ctypes.data, ctypes.byref(warn))),
            ("crossprod(x, y[1e4 x 16])", lambda L, p: getattr(L, p + "crossprod2_SVT_mat")(ctypes.addressof(view), y.ctypes.data, 10_000, 16, 14, 0, o2.ctypes.data)),
            ("crossprod(svt1) 25000 x 400 @ 7 %", lambda L, p: getattr(L, p + "crossprod1_SVT")(ctypes.addressof(v1w), o3.ctypes.data))]
    for name, f in rows:
        print(f"{tag} {name:36s} libsvt_hip {best(lambda: f(hl, 'svt_')):8.3f} ms   cpu oracle {best(lambda: f(orc, 'orc_')):8.3f} ms", flush=True)


table("config 1 (1e4 x 1e3 @ 1 %):")
