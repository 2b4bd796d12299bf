"""Child process of tests/test_rccl_one_rank.py: ONE rank, backend `nccl` (= RCCL on ROCm), on GPU 0.  A one-GPU box
cannot measure scaling; what it can show is that RCCL loads beside libsvt_hip.so (with HSA_ENABLE_IPC_MODE_LEGACY=0,
the setting the N > 1 runs need), and that the collective code path of sparsearray_amd/parallel.py -- the asynchronous
all-reduce of `ShardedCrossprod` on its own stream with two result buffers, the all-reduce of `sharded_colsums_rows`,
the all-gathers of the leaf-sharded statistics -- runs against the real backend and leaves the results it leaves
without a group, bit for bit (an all-reduce / all-gather over one rank is the identity).  Writes a JSON verdict to
argv[1]."""
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["LOCAL_RANK"] = "0"
for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "30477")):
    os.environ.setdefault(k_, v_)           # (started by hand, e.g. from tools/debug/r6_profiles.sh)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    out_path = sys.argv[1]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from sparsearray_amd import parallel as par
    from sparsearray_amd import synth
    from sparsearray_amd.device import DeviceCSC, colstats, rowsum
    nrow, ncol, dens, K = 256_000, 10_000, 0.01, 128
    cp, ri, v = synth.random_device_csc(nrow, ncol, dens, seed=11, device=dev)
    A = DeviceCSC(nrow, cp, ri, v)
    Y = synth.random_dense(nrow, K, seed=111, device=dev)
    grp = torch.randint(1, 101, (nrow,), device=dev, dtype=torch.int32)
    # without a group: no collective anywhere
    sc0 = par.ShardedCrossprod(A, K)
    for _ in range(3):
        sc0.step(Y)
    r0 = sc0.result().clone()
    c0 = par.sharded_colsums_rows(A).clone()
    v0 = par.sharded_colstats(A, [(0, ncol)], "var1").clone()
    g0 = par.sharded_rowsum(A, [(0, ncol)], grp, 100).clone()
    torch.cuda.synchronize()
    # the group of one rank over RCCL, every collective forced
    dist.init_process_group("nccl", device_id=dev)
    par.force_collectives(True)
    sc1 = par.ShardedCrossprod(A, K)
    assert len(sc1.outs) == 2                        # two result buffers, as with N > 1
    for _ in range(5):                               # both buffers reused: every step waits for its buffer's all-reduce
        sc1.step(Y)
    pending = sum(w is not None for w in sc1.pending)
    r1 = sc1.result().clone()
    c1 = par.sharded_colsums_rows(A).clone()
    v1 = par.sharded_colstats(A, [(0, ncol)], "var1").clone()
    g1 = par.sharded_rowsum(A, [(0, ncol)], grp, 100).clone()
    t = torch.ones(4, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    torch.cuda.synchronize()
    verdict = {
        "backend": dist.get_backend(), "world_size": dist.get_world_size(),
        "rccl_version": list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
        "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
        "all_reduces_in_flight_after_5_steps": pending,
        "crossprod_identical": bool(torch.equal(r0, r1)), "colsums_identical": bool(torch.equal(c0, c1)),
        "colvars_identical": bool(torch.equal(v0, v1)), "rowsum_identical": bool(torch.equal(g0, g1)),
        "allreduce_of_ones": t.tolist(),
    }
    dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(verdict, f)


if __name__ == "__main__":
    main()
