"""BASELINE config 2b: A %*% Y (A 1e6 x 1e4 @ 1 %, Y 1e4 x 128) = crossprod(t(A), Y) on the layout of t(A): one launch
against one launch per round of workgroups (2), and per round with the last round cut by rows (1, the default)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan, set_round_launches
dev = torch.device("cuda", 0)
N, M, K = 1_000_000, 10_000, 128
cp, ri, v = synth.random_device_csc(N, M, 0.01, seed=1, device=dev)
A = DeviceCSC(N, cp, ri, v)
T = A.t()
plan = PbcPlan(T, K)
Y = synth.random_dense(M, K, seed=202, device=dev)
outs = []
for on in (2, 1, 0, 2, 1):
    set_round_launches(on)
    out = torch.empty((K, N), dtype=torch.float64, device=dev)
    for _ in range(5):
        plan.run(Y, M, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.run(Y, M, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"round launches {on}: {ms:.3f} ms  {A.nnz / ms / 1e6:.1f} GNZ/s  frac {(A.nnz * 12 + M * K * 8 + N * K * 8) / ms / 1e6 / 8000:.3f}", flush=True)
    outs.append(out)
print("same result (whole last round vs one launch):", bool(torch.equal(outs[0], outs[2])))
d = (outs[0] - outs[1]).abs().max().item()
print("last round cut by rows vs whole: max abs diff", d, "relative", d / outs[0].abs().max().item())
set_round_launches(True)
