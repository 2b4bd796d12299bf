"""One rank's share of BASELINE config 4 through the XCD-paced gather kernel with part of the CUs left
idle (svt_dev_pbc_set_spare_cus): does the time follow the bytes in flight (latency-bound) or stay
(L2 throughput-bound)?  usage: config4_occupancy.py [spare,...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan, set_gather_pacing, set_spare_cus

spares = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 64, 128, 192]
nrow, ncol, K = 1_250_000, 50_000, 128
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.001, seed=4, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
Y = synth.random_dense(nrow, K, seed=104, device=dev)
out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


plan = PbcPlan(A, K, 40, 4, 10)
set_gather_pacing()
for sp in spares:
    set_spare_cus(sp)
    ms = timed(lambda: plan.run(Y, nrow, out))
    print(f"spare CUs {sp:3d}: {ms:7.3f} ms", flush=True)
set_spare_cus(0)
