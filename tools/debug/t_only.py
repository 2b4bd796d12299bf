import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(1_000_000, 10_000, 0.01, seed=1, device=dev)
A = DeviceCSC(1_000_000, cp, ri, v)
for _ in range(3):
    T = A.t(); torch.cuda.synchronize(); del T
