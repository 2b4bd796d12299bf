#!/usr/bin/env python3
"""Tuning harness for the PBC crossprod kernel (run on the GPU box):
times the kernel for several (CBW, WPB, logR) and, with --ablate, the
timing-only builds (no staging / no record loop)."""
import argparse
import os
import sys
import time

import torch

os.environ["SVT_HIP_TUNING"] = "1"   # needs `make -C sparsearray_amd/csrc TUNING=1` (svt_dev_pbc_set_debug)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsearray_amd import synth  # noqa: E402
from sparsearray_amd.device import DeviceCSC, PbcPlan, _lib  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--nrow", type=int, default=1_000_000)
p.add_argument("--ncol", type=int, default=10_000)
p.add_argument("--density", type=float, default=0.01)
p.add_argument("--K", type=int, default=128)
p.add_argument("--cfgs", default="32,16,8")
p.add_argument("--ablate", action="store_true")
p.add_argument("--prof", action="store_true", help="per-section cycle counts of workgroup 0")
p.add_argument("--reps", type=int, default=5)
p.add_argument("--nsplits", default="0")
p.add_argument("--staggers", default="7")
p.add_argument("--aheads", default="20", help="record-touch look-ahead, tenths of a tile")
p.add_argument("--ldy0", action="store_true", help="timing experiment: all dense columns alias column 0 (8 MB, cache resident)")
a = p.parse_args()

dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(a.nrow, a.ncol, a.density, seed=1, device=dev)
Y = synth.random_dense(a.nrow, a.K, seed=101, device=dev)
A = DeviceCSC(a.nrow, cp, ri, v)
out = torch.zeros((a.K, a.ncol), dtype=torch.float64, device=dev)
lib = _lib()


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for cfg in a.cfgs.split(";"):
    cbw, wpb, logr = (int(x) for x in cfg.split(","))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan = PbcPlan(A, a.K, cbw, wpb, logr)
    torch.cuda.synchronize(); build = (time.perf_counter() - t0) * 1e3
    row = [f"cfg cbw={cbw} wpb={wpb} logR={logr}: build {build:.1f} ms"]
    for ns, stg, ah in ((int(x), int(y), int(z)) for x in a.nsplits.split(",") for y in a.staggers.split(",")
                        for z in a.aheads.split(",")):
        lib.svt_dev_pbc_set_debug(100 + ns)
        lib.svt_dev_pbc_set_debug(300 + ah)
        lib.svt_dev_pbc_set_debug(200 + stg)
        del plan
        plan = PbcPlan(A, a.K, cbw, wpb, logr)      # workspace depends on the split count
        for mode, name in ((0, "full"), (2, "no-compute")):
            if mode and not a.ablate:
                continue
            lib.svt_dev_pbc_set_debug(mode)
            ms = timed(lambda: plan.run(Y, 0 if a.ldy0 else a.nrow, out), a.reps)
            row.append(f"[nsplit {ns} stagger {stg} ahead {ah}] {name} {ms:.3f} ms ({A.nnz / ms / 1e6:.1f} GNZ/s)")
        if a.prof:
            import ctypes
            lib.svt_dev_pbc_set_debug(3)
            ms = timed(lambda: plan.run(Y, a.nrow, out), 2)
            buf = (ctypes.c_ulonglong * 128)()
            lib.svt_dev_pbc_read_prof.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            assert lib.svt_dev_pbc_read_prof(plan.ws.data_ptr(), buf) == 0
            row.append(f"\n  prof build {ms:.3f} ms; per wavefront cycles "
                       "[fetch|records, records|dma-wait, barrier1|barrier, commit|issue, barrier2|prescan, panels] (register-staged|DMA kernel):")
            for w in range(wpb):
                row.append("\n    w%02d " % w + " ".join("%9d" % buf[w * 8 + i] for i in range(8)))
        lib.svt_dev_pbc_set_debug(0)
    lib.svt_dev_pbc_set_debug(100)
    print("  ".join(row), flush=True)
    del plan
