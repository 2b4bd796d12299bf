"""One rank's share of BASELINE config 4 (1.25e6 x 5e4 @ 0.1 %, Y 1.25e6 x 128) on one GPU: the gather
product with the unpaced kernels (one launch per row chunk) and with the XCD-paced persistent kernel
(crossprod_pbc_gatherx_kernel) over a sweep of its pacing knobs and panel heights.
usage: config4_pacing.py [nrow] [logR,...] ["dsync:spin,..."]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan, set_gather_pacing

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
logrs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [9, 10]
sweep = sys.argv[3] if len(sys.argv) > 3 else "-1:256,1:256,2:256,3:256,2:16,2:4096,1000000:1"
ncol, K = int(os.environ.get("C4_NCOL", 50_000)), 128
dens = float(os.environ.get("C4_DENSITY", 0.001))
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, dens, seed=4, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
Y = synth.random_dense(nrow, K, seed=104, device=dev)
out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
alg = A.nnz * 12 + nrow * K * 8 + ncol * K * 8


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# reference: 40 sampled leaves by plain torch gathers
g = torch.Generator().manual_seed(2)
cols = torch.randint(0, ncol, (40,), generator=g).tolist() + [0, ncol - 1]
want = {}
for c in cols:
    lo, hi = int(A.col_ptr[c]), int(A.col_ptr[c + 1])
    rows = A.row_idx[lo:hi].long()
    want[c] = (Y[:, rows] * A.val[lo:hi]).sum(dim=1)

for logr in logrs:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan = PbcPlan(A, K, 40, 4, logr)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) * 1e3
    print(f"layout (40, 4, {logr}): {tb:.1f} ms", flush=True)
    for item in sweep.split(","):
        d, sp = (int(x) for x in item.split(":"))
        set_gather_pacing(d, sp)
        out.zero_()
        ms = timed(lambda: plan.run(Y, nrow, out))
        worst = max(float((out[:, c] - want[c]).abs().max() / want[c].abs().max()) for c in cols)
        print(f"  logR {logr} dsync {d:8d} spin {sp}: {ms:7.3f} ms  {A.nnz / ms / 1e6:6.1f} GNZ/s  gathered {A.nnz * K * 8 / ms / 1e9:5.1f} TB/s  "
              f"frac {alg / ms / 1e6 / 8000:.4f}  max rel err {worst:.2e}", flush=True)
    del plan
set_gather_pacing()
