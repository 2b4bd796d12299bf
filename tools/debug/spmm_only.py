import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, matmul_csc_csc, _lib
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(1_000_000, 10_000, 0.01, seed=1, device=dev)
A = DeviceCSC(1_000_000, cp, ri, v)
bcp, bri, bv = synth.random_device_csc(10_000, 128, 0.01, seed=303, device=dev)
B = DeviceCSC(10_000, bcp, bri, bv)
out = torch.empty((128, 1_000_000), dtype=torch.float64, device=dev)
ws = torch.empty(_lib().svt_dev_matmul_csc_csc_ws_bytes(A.handle), dtype=torch.uint8, device=dev)
for _ in range(4):
    matmul_csc_csc(A, B, out=out, ws=ws); torch.cuda.synchronize()
