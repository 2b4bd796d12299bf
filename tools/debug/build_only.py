import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(1_000_000, 10_000, 0.01, seed=1, device=dev)
A = DeviceCSC(1_000_000, cp, ri, v)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = PbcPlan(A, 128, 0, 0, 0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("build %d: %.3f ms" % (i, (t1 - t0) * 1e3))
    del p
