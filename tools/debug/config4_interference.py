"""One rank's share of BASELINE config 4 while ANOTHER stream keeps a few CUs busy at the moment the product is
launched (what a collective's kernels do at N > 1): does the paced gather kernel keep its pace when some of its
workgroups can only start late?  usage: config4_interference.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan

nrow, ncol, K = 1_250_000, 50_000, 128
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.001, seed=4, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
Y = synth.random_dense(nrow, K, seed=104, device=dev)
out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
plan = PbcPlan(A, K, 0, 0, 0)
side = torch.cuda.Stream()
# the "collective": 51 MB summed a few times on the side stream (same bytes as the all-reduce of the result)
buf = torch.zeros((8, K, ncol), dtype=torch.float64, device=dev)
acc = torch.zeros((K, ncol), dtype=torch.float64, device=dev)


def step(with_side):
    if with_side:
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            for _ in range(3):
                torch.sum(buf, dim=0, out=acc)
    plan.run(Y, nrow, out)


for with_side in (False, True, False, True):
    step(with_side); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        step(with_side)
    e1.record(); torch.cuda.synchronize()
    print(f"side stream busy at launch: {with_side}:  {e0.elapsed_time(e1) / 5:.3f} ms per product", flush=True)
