"""Row splits beyond a multiple of 8 (kernels_mult_pbc.hip, pick_nsplit): at 14 column blocks x 2 dense
tiles (ncol = 8960, CBW 40) eight splits use 224 CUs; a ninth, dealt over all XCDs, uses the other 32.
Tuning build (forces the split count); config 2a otherwise."""
import os, sys
os.environ["SVT_HIP_TUNING"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth, _hip
from sparsearray_amd.device import DeviceCSC, PbcPlan
lib = _hip.init()
dev = torch.device("cuda", 0)
nrow, K = 1_000_000, 128
for ncol in (8960, 10_000):
    cp, ri, v = synth.random_device_csc(nrow, ncol, 0.01, seed=1, device=dev)
    A = DeviceCSC(nrow, cp, ri, v)
    Y = synth.random_dense(nrow, K, seed=101, device=dev)
    out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
    ref = None
    for ns in (8, 9, 10, 0):
        lib.svt_dev_pbc_set_debug(100 + ns)
        plan = PbcPlan(A, K, 40, 16, 7)
        for _ in range(3): plan.run(Y, nrow, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.run(Y, nrow, out)
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f"ncol {ncol} forced splits {ns}: {e0.elapsed_time(e1) / 20:.3f} ms per product (max rel diff vs 8 splits {err:.1e})", flush=True)
        del plan
    lib.svt_dev_pbc_set_debug(100)
    del A, cp, ri, v, Y, out
