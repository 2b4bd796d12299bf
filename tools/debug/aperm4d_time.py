"""aperm of a 4-d array (2e4 x 2e3 x 10 x 64 @ 0.5 %: BASELINE config 5's nonzeros with the second axis split) for
permutations that are none of the special forms: round 5 composes them (leaf-preserving step, first two axes swapped,
leaf-preserving step); device time per call (events) -- under tools/debug/prof_py.sh for the per-kernel split."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC
D = (20_000, 2_000, 10, 64)
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(D[0], D[1] * D[2] * D[3], 0.005, seed=5, device=dev)
A = DeviceCSC(D[0], cp, ri, v)
for perm in [(2, 4, 1, 3), (3, 2, 4, 1), (4, 3, 2, 1), (2, 1, 3, 4), (1, 4, 2, 3)]:
    for _ in range(2):
        P, pdim = A.aperm(D, perm); del P
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        P, pdim = A.aperm(D, perm); del P
    e1.record(); torch.cuda.synchronize()
    print(f"aperm {perm} of {D}: {e0.elapsed_time(e1) / 5:.3f} ms per call  ({A.nnz} nonzeros)", flush=True)
