"""aperm at BASELINE config 5 (2e4 x 2e4 x 64 @ 0.5 %): c(1,3,2) (leaf-preserving) and c(3,1,2) (slab form), device time
per call (events) -- run under tools/debug/prof_py.sh for the per-kernel split."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC
D = (20_000, 20_000, 64)
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(D[0], D[1] * D[2], 0.005, seed=5, device=dev)
A = DeviceCSC(D[0], cp, ri, v)
perms = [tuple(int(c) for c in a) for a in sys.argv[1:]] or [(1, 3, 2), (3, 1, 2), (2, 1, 3), (2, 3, 1), (3, 2, 1)]
for perm in perms:
    for _ in range(2):
        P, pdim = A.aperm(D, perm); del P
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        P, pdim = A.aperm(D, perm); del P
    e1.record(); torch.cuda.synchronize()
    print(f"aperm {perm}: {e0.elapsed_time(e1) / 5:.3f} ms per call", flush=True)
