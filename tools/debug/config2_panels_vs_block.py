import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, crossprod_csc_csc, set_sparse_crossprod_panel, _lib
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(1_000_000, 10_000, 0.01, seed=1, device=dev)
A = DeviceCSC(1_000_000, cp, ri, v); At = A.t()
out = torch.empty((10_000, 10_000), dtype=torch.float64, device=dev)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ref = None
for one, ps in ((-1, -1), (0, 13), (0, 12), (0, 11), (0, 14)):
    set_sparse_crossprod_panel(one, ps)
    ws = torch.empty(_lib().svt_dev_crossprod_csc_csc_ws_bytes(At.handle), dtype=torch.uint8, device=dev)
    for sym in (True, False):
        ms = timed(lambda: crossprod_csc_csc(At, A, sym=sym, out=out, ws=ws))
        if ref is None: ref = out.clone()
        print(f"one_block_max {one} log2_panel {ps} sym {sym}: {ms:.3f} ms  max|diff| {float((out-ref).abs().max()):.2e}", flush=True)
set_sparse_crossprod_panel(-1, -1)
