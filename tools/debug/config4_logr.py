"""Gather kernel at BASELINE config 4 with taller panels (layout (40, 4, logR), tuning build: SVT_PBG_CHUNK = panels per
row split and launch):  SVT_PBG_CHUNK=80 python tools/debug/config4_logr.py 11"""
import os, sys, time
os.environ["SVT_HIP_TUNING"] = "1"
import torch
sys.path.insert(0, os.getcwd())
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan
nrow, ncol, K = 10_000_000, 50_000, 128
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.001, seed=4, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
Y = synth.random_dense(nrow, K, seed=104, device=dev)
out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
logr = int(sys.argv[1])
plan = PbcPlan(A, K, 40, 4, logr)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: plan.run(Y, nrow, out))
print(f"logR {logr} chunk {os.environ.get('SVT_PBG_CHUNK')}: {ms:.2f} ms  checksum {float(out.abs().sum()):.6e}", flush=True)
