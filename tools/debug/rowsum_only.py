import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, rowsum, rowsums, _lib
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(1_000_000, 10_000, 0.01, seed=1, device=dev)
A = DeviceCSC(1_000_000, cp, ri, v)
grp = torch.randint(1, 1001, (1_000_000,), device=dev, dtype=torch.int32)
rs_out = torch.empty(1_000_000, dtype=torch.float64, device=dev)
rs_ws = torch.empty(_lib().svt_dev_rowstats_ws_bytes(1_000_000, 10_000), dtype=torch.uint8, device=dev)
for _ in range(4):
    o = rowsum(A, grp, 1000); rowsums(A, out=rs_out, ws=rs_ws); torch.cuda.synchronize()
