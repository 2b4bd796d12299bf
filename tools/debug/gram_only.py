"""crossprod(A) at BASELINE config-2 scale through the sparse-aware kernel only (for rocprofv3: kernel trace / --pmc).
usage: gram_only.py [reps] [sym]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, crossprod_csc_csc, _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sym = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) != 0
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(1_000_000, 10_000, 0.01, seed=1, device=dev)
A = DeviceCSC(1_000_000, cp, ri, v)
At = A.t()
out = torch.empty((10_000, 10_000), dtype=torch.float64, device=dev)
ws = torch.empty(_lib().svt_dev_crossprod_csc_csc_ws_bytes(At.handle), dtype=torch.uint8, device=dev)
for _ in range(reps):
    crossprod_csc_csc(At, A, sym=sym, out=out, ws=ws)
torch.cuda.synchronize()
print("done", float(out[0, 0]))
