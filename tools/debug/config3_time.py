"""BASELINE config 3 at full size on the GPU box: A %*% B (A 1e6 x 1e4 @ 1 %, B 1e4 x 128 sparse
@ 1 %) through the one-call host entry point (svt_matmul_SVT_SVT: upload, device transposition,
layout build, product, download), its device-level pieces, and rowsum() with 1e3 groups."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth, _hip
from sparsearray_amd.svt import make_view_from_csc
from sparsearray_amd.device import DeviceCSC, PbcPlan, rowsum
lib = _hip.init()
nrow, ncol, K = 1_000_000, 10_000, 128
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.01, seed=1, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
bcp, bri, bv = synth.random_device_csc(ncol, K, 0.01, seed=2, device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# device-level pieces
print(f"t(A) on the device                {timed(lambda: A.t(), 3):8.2f} ms")
At = A.t()
t0 = time.perf_counter(); plan = PbcPlan(At, K); torch.cuda.synchronize()
print(f"panel-blocked layout of t(A)      {(time.perf_counter() - t0) * 1e3:8.2f} ms")
Bd = torch.zeros(K, ncol, dtype=torch.float64, device=dev)      # dense B, column-major (K columns of ncol)
cols = torch.repeat_interleave(torch.arange(K, device=dev), (bcp[1:] - bcp[:-1]))
Bd[cols, bri.long()] = bv
out = torch.empty(K, nrow, dtype=torch.float64, device=dev)      # column-major nrow x K
ms = timed(lambda: plan.run(Bd, ncol, out, 1, nrow))
print(f"product kernel (t(A) layout, dense B) {ms:8.3f} ms  {A.nnz / ms / 1e6:6.1f} GNZ/s")
grp = torch.randint(0, 1000, (nrow,), dtype=torch.int32, device=dev)
rs = torch.empty(ncol, 1000, dtype=torch.float64, device=dev)
ms = timed(lambda: rowsum(A, grp, 1000, out=rs))
print(f"rowsum(A, 1e3 groups)             {ms:8.3f} ms  {A.nnz / ms / 1e6:6.1f} GNZ/s  "
      f"{(A.nnz * 12 + nrow * 4 + ncol * 8000) / ms / 1e6:6.0f} GB/s (algorithmic)")
ref = out.clone()
del plan, At, out, rs
torch.cuda.empty_cache()

# host level, one call
cph, rih, vh = cp.cpu().numpy(), ri.cpu().numpy(), v.cpu().numpy()
xv = make_view_from_csc((nrow, ncol), "double", cph, rih, vh)
yv = make_view_from_csc((ncol, K), "double", bcp.cpu().numpy(), bri.cpu().numpy(), bv.cpu().numpy())
res = np.zeros((nrow, K), order="F")
fn = lib.svt_matmul_SVT_SVT
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
for rep in range(3):
    t0 = time.perf_counter()
    rc = fn(ctypes.addressof(xv), ctypes.addressof(yv), res.ctypes.data)
    dt = time.perf_counter() - t0
    print(f"svt_matmul_SVT_SVT host level: rc={rc} {dt * 1e3:.1f} ms  ({len(rih) / dt / 1e9:.2f} GNZ/s)")
err = np.abs(res.T - ref.cpu().numpy()).max()
print("max |host-level - device-level| =", err)
