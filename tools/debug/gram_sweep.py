"""Knobs of the sparse-aware crossprod kernel at BASELINE config-2 scale (tuning build: SVT_HIP_TUNING=1).
usage: SVT_HIP_TUNING=1 gram_sweep.py [nrow ncol density reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, crossprod_csc_csc, _lib

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, dens, seed=1, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
At = A.t()
out = torch.empty((ncol, ncol), dtype=torch.float64, device=dev)
ws = torch.empty(_lib().svt_dev_crossprod_csc_csc_ws_bytes(At.handle), dtype=torch.uint8, device=dev)


def timed(fn, n=reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
KEYS = ("SVT_GRAM_SYMK", "SVT_GRAM_G", "SVT_GRAM_NT", "SVT_GRAM_SU")
CASES = [(True, {}), (True, {"SVT_GRAM_SU": "2"}), (True, {"SVT_GRAM_SU": "4"}), (True, {"SVT_GRAM_G": "8"}), (True, {"SVT_GRAM_G": "32"}),
         (True, {"SVT_GRAM_NT": "512"}), (True, {"SVT_GRAM_SYMK": "0"}),
         (False, {}), (False, {"SVT_GRAM_SYMK": "0"})]
for sym, env in CASES:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    ms = timed(lambda: crossprod_csc_csc(At, A, sym=sym, out=out, ws=ws))
    if ref is None:
        ref = out.clone()
    d = float((out - ref).abs().max())
    print(f"sym={int(sym)} {str(env):50s} {ms:8.3f} ms   max |diff to first| {d:.2e}", flush=True)
