#!/bin/bash
# tuning sweep of the sparse x sparse row-panel kernel (env overrides exist in tuning builds only)
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "13 1 32" "13 1 16" "14 1 32" "14 1 64" "12 1 16" "13 1 8"; do
  set -- $cfg
  SVT_SPMM_PS=$1 SVT_SPMM_KW=$2 SVT_SPMM_G=$3 python - <<PY
import os, sys, time, torch
sys.path.insert(0, ".")
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, matmul_csc_csc, _lib
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(1_000_000, 10_000, 0.01, seed=1, device=dev)
A = DeviceCSC(1_000_000, cp, ri, v)
bcp, bri, bv = synth.random_device_csc(10_000, 128, 0.01, seed=303, device=dev)
B = DeviceCSC(10_000, bcp, bri, bv)
out = torch.empty((128, 1_000_000), dtype=torch.float64, device=dev)
ws = torch.empty(_lib().svt_dev_matmul_csc_csc_ws_bytes(A.handle) * 8, dtype=torch.uint8, device=dev)
matmul_csc_csc(A, B, out=out, ws=ws); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): matmul_csc_csc(A, B, out=out, ws=ws)
e1.record(); torch.cuda.synchronize()
print("ps $1 KW $2 G $3: %.3f ms  checksum %.6f" % (e0.elapsed_time(e1) / 5, float(out.sum().item())))
PY
done
