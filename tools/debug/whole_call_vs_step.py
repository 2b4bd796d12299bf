"""Why `extras.crossprod_whole_call.ms` (1.883) and `ms_per_step` (1.787) differed in BENCH_r04 for the same two phases:
the same product timed (a) as the bench's steps, (b) as one svt_dev_crossprod_pbc call, in alternation, short and long runs."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth, parallel as par
from sparsearray_amd.device import DeviceCSC
dev = torch.device("cuda", 0)
nrow, ncol, K = 1_000_000, 10_000, 128
cp, ri, v, _ = synth.random_device_csc_blocked(nrow, ncol, 0.01, seed=1, device=dev, nblocks=8, first=0, last=8)
Y = synth.random_dense_blocked(nrow, K, seed=101, device=dev, nblocks=8, first=0, last=8)
A = DeviceCSC(nrow, cp, ri, v)
sc = par.ShardedCrossprod(A, K)
outx = torch.zeros((K, ncol), dtype=torch.float64, device=dev)


def wall(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(3):
    for reps in (10, 20, 100, 400):
        a = wall(lambda: sc.step(Y), reps)
        b = wall(lambda: sc.plan.run(Y, nrow, outx), reps)
        c = ev(lambda: sc.plan.run(Y, nrow, outx), reps)
        d = ev(lambda: sc.step(Y), reps)
        print(f"round {rnd} reps {reps}: step wall {a:.4f}  whole call wall {b:.4f}  whole call events {c:.4f}  step events {d:.4f}", flush=True)

# the GPU's clocks after an idle gap: 10 whole calls timed right after `gap` seconds of nothing
for gap in (0.0, 0.01, 0.1, 1.0, 0.0):
    torch.cuda.synchronize(); time.sleep(gap)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        sc.plan.run(Y, nrow, outx)
    e1.record(); torch.cuda.synchronize()
    first10 = e0.elapsed_time(e1) / 10
    e0.record()
    for _ in range(10):
        sc.plan.run(Y, nrow, outx)
    e1.record(); torch.cuda.synchronize()
    print(f"idle {gap} s, then 10 calls: {first10:.4f} ms per call; the next 10: {e0.elapsed_time(e1) / 10:.4f}", flush=True)
