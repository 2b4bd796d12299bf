"""Host-level cost at BASELINE config 2: svt_upload() (marshal + H2D) and the whole
C_crossprod2_SVT_mat call (run on the GPU box)."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth, _hip
from sparsearray_amd.svt import make_view_from_csc
lib = _hip.init()
nrow, ncol, K = 1_000_000, 10_000, 128
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.01, seed=1, device=dev)
cp, ri, v = cp.cpu().numpy(), ri.cpu().numpy(), v.cpu().numpy()
view = make_view_from_csc((nrow, ncol), "double", cp, ri, v)
lib.svt_upload.restype = ctypes.c_void_p
lib.svt_upload.argtypes = [ctypes.c_void_p]
lib.svt_release.argtypes = [ctypes.c_void_p]
for rep in range(3):
    t0 = time.perf_counter()
    h = lib.svt_upload(ctypes.addressof(view))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"svt_upload: {dt*1e3:.1f} ms  ({(len(ri)*12)/dt/1e9:.1f} GB/s)")
    lib.svt_release(h)
y = np.asfortranarray(np.random.default_rng(2).uniform(-1, 1, (nrow, K)))
out = np.zeros((ncol, K), order="F")
fn = lib.svt_crossprod2_SVT_mat
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for rep in range(2):
    t0 = time.perf_counter()
    rc = fn(ctypes.addressof(view), y.ctypes.data, nrow, K, 14, 0, out.ctypes.data)
    dt = time.perf_counter() - t0
    print(f"C_crossprod2_SVT_mat host level: rc={rc} {dt*1e3:.1f} ms  ({len(ri)/dt/1e9:.2f} GNZ/s)")

# resident operands: the same calls with the device copy (and its panel-blocked layout) kept
lib.svt_resident_set_limit.argtypes = [ctypes.c_size_t]
lib.svt_resident_set_limit(8 << 30)
for rep in range(3):
    t0 = time.perf_counter()
    rc = fn(ctypes.addressof(view), y.ctypes.data, nrow, K, 14, 0, out.ctypes.data)
    dt = time.perf_counter() - t0
    print(f"C_crossprod2_SVT_mat, resident x: rc={rc} {dt*1e3:.1f} ms  ({len(ri)/dt/1e9:.2f} GNZ/s)")
from sparsearray_amd._dispatch import CAbiDispatcher
op_sum = CAbiDispatcher(lib, "svt_")._opcode("sum")
cs = lib.svt_colStats_SVT
cs.restype = ctypes.c_int
cs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
res = np.zeros(ncol)
warn = ctypes.c_int(0)
for lim in (8 << 30, 0):
    lib.svt_resident_set_limit(lim)
    for rep in range(3):
        t0 = time.perf_counter()
        rc = cs(ctypes.addressof(view), op_sum, 0, float("nan"), 1, res.ctypes.data, ctypes.addressof(warn))
        dt = time.perf_counter() - t0
        print(f"C_colStats_SVT(sum), resident limit {lim >> 30} GiB: rc={rc} {dt*1e3:.2f} ms")
