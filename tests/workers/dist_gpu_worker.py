"""Child process of tests/test_hip_configs.py::test_two_ranks_same_device_*: one rank of a 2-rank
`gloo` job whose ranks both compute on GPU 0 with the HIP library (the one-GPU rehearsal of the
multi-GPU path; the RCCL run on 8 GPUs is the driver's).  Runs the sharded operations of
sparsearray_amd/parallel.py on a slice of BASELINE configs 2a and 4 and compares every reduced /
gathered result with the one-rank result computed from the unsharded operand.  Rank 0 writes a JSON
verdict to argv[1]."""
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["LOCAL_RANK"] = "0"                       # both ranks bind the HIP library to GPU 0
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    from sparsearray_amd import parallel as par
    from sparsearray_amd import synth
    from sparsearray_amd.device import DeviceCSC, PbcPlan, colstats, rowsum
    verdict = {}
    for name, (nrow, ncol, dens, K) in {
        "config2a_slice": (256_000, 10_000, 0.01, 128),       # 2.56e7 nonzeros
        "config4_slice": (1_280_000, 50_000, 0.001, 64),      # 6.4e7 nonzeros, 5-record tiles
    }.items():
        # the global operand, identical on both ranks (same seeds), and this rank's row blocks
        cp, ri, v, _ = synth.random_device_csc_blocked(nrow, ncol, dens, seed=11, device=dev)
        Y = synth.random_dense_blocked(nrow, K, seed=111, device=dev)
        A = DeviceCSC(nrow, cp, ri, v)
        per = 8 // world
        cpl, ril, vl, (r0, r1) = synth.random_device_csc_blocked(nrow, ncol, dens, seed=11, device=dev,
                                                                 first=rank * per, last=(rank + 1) * per)
        Al = DeviceCSC(r1 - r0, cpl, ril, vl)
        Yl = Y[:, r0:r1].contiguous()
        # the device-side row filter gives the same shard as generating the blocks directly
        As, (s0, s1) = par.shard_rows_device(A, rank, world, align=128)
        same_shard = (s0, s1) == (r0, r1) and torch.equal(As.col_ptr, Al.col_ptr) and \
            torch.equal(As.row_idx, Al.row_idx) and torch.equal(As.val, Al.val)
        # crossprod: row-sharded product + all-reduce vs the one-rank product
        sc = par.ShardedCrossprod(Al, K)
        sc.step(Yl)
        sc.step(Yl)                                       # second step: the other buffer, waits on the first
        got = sc.result().clone()
        ref = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
        PbcPlan(A, K).run(Y, nrow, ref)
        torch.cuda.synchronize()
        scale = float(ref.abs().max())
        err_cp = float((got - ref).abs().max()) / scale
        # the same through the peer-copy reducer (IPC windows, interprocess events; here both ranks on one device)
        sp = par.ShardedCrossprod(Al, K, reducer="peer")
        for _ in range(5):                                # five steps: both buffers reused twice
            sp.step(Yl)
        gotp = sp.result().clone()
        torch.cuda.synchronize()
        err_peer = float((gotp - ref).abs().max()) / scale
        dist.barrier()
        del sp
        # colSums of the row shards, all-reduced, vs the one-rank colSums
        cs = par.sharded_colsums_rows(Al)
        cs_ref, _ = colstats(A, "sum")
        err_cs = float((cs - cs_ref).abs().max()) / float(cs_ref.abs().max())
        # leaf-sharded colVars (gathered scalars) and rowsum (gathered slabs): bit-identical
        Ac, blocks = par.shard_cols_device(A, rank, world)
        cv = par.sharded_colstats(Ac, blocks, "var1")
        cv_ref, _ = colstats(A, "var1")
        ng = 1000
        g = torch.Generator(device=dev); g.manual_seed(5)
        grp = torch.randint(1, ng + 1, (nrow,), generator=g, device=dev, dtype=torch.int32)
        rs = par.sharded_rowsum(Ac, blocks, grp, ng)
        rs_ref = rowsum(A, grp, ng)
        torch.cuda.synchronize()
        err_rs = float((rs - rs_ref).abs().max()) / max(float(rs_ref.abs().max()), 1e-300)
        # unary crossprod(A) with the LEAVES sharded (last but one row): every rank all-gathers the
        # other operand and forms its own rows of the ncol x ncol result with the sparse-aware kernel; against the one-rank
        # symmetric product (config2a slice only: 10^4 columns -- the config-4 slice's result would be 20 GB)
        err_sp = 0.0
        if ncol <= 10_000:
            from sparsearray_amd.device import crossprod_csc_csc
            lp = par.device_sparse_crossprod_block(Ac)
            c0, c1 = blocks[rank]
            k0, k1 = int(cp[c0]), int(cp[c1])
            got_sp = par.sharded_crossprod_sparse(lp, ((cp[c0:c1 + 1] - k0).contiguous(), ri[k0:k1], v[k0:k1]), blocks)
            ref_sp, flag_sp = crossprod_csc_csc(A.t(), A, sym=True)
            torch.cuda.synchronize()
            assert int(flag_sp.item()) == 0 and int(lp.flag.item()) == 0
            err_sp = float((got_sp - ref_sp.t()).abs().max()) / float(ref_sp.abs().max())
            del got_sp, ref_sp, lp
        verdict[name] = {"same_shard": bool(same_shard), "crossprod_rel_err": err_cp, "crossprod_peer_rel_err": err_peer,
                         "sparse_crossprod_rel_err": err_sp,
                         "colsums_rel_err": err_cs,
                         "colvars_identical": bool(torch.equal(cv, cv_ref)), "rowsum_rel_err": err_rs,
                         "nnz": A.nnz, "rows_rank": r1 - r0, "blocks": blocks}
        del sc, A, Al, As, Ac, Y, Yl, got, ref
        torch.cuda.empty_cache()
    # ---- BASELINE config 5, a slice: 3-d array cut along axis 1; column statistics per leaf, rowSums over
    # axis 2 (every output cell owned by one rank), aperm of the shard; results gathered and compared
    from sparsearray_amd.device import rowsums
    D = (4000, 600, 16)
    cp, ri, v = synth.random_device_csc(D[0], D[1] * D[2], 0.005, seed=21, device=dev)
    A = DeviceCSC(D[0], cp, ri, v)
    blocks5 = [par.row_block(D[1], r, world) for r in range(world)]
    As, sdim, (lo, hi) = par.shard_axis_device(A, D, 1, rank, world)
    cs_loc, _ = colstats(As, "sum")                                   # one sum per leaf of the shard: (hi-lo) x 16
    cs = par.gather_axis(cs_loc, (D[1], D[2]), 0, blocks5)
    cs_ref, _ = colstats(A, "sum")
    rs_loc = rowsums(As, inner=sdim[1])                               # D[0] x (hi-lo): sum over axis 2
    rs = par.gather_axis(rs_loc, (D[0], D[1]), 1, blocks5)
    rs_ref = rowsums(A, inner=D[1])
    Ps, pdim = As.aperm(sdim, (1, 3, 2))                              # leaf-preserving on the shard
    cs2_loc, _ = colstats(Ps, "sum")                                  # leaves in (axis 2, axis 1) order
    cs2 = par.gather_axis(cs2_loc, (D[2], D[1]), 1, blocks5)
    Pf, _ = A.aperm(D, (1, 3, 2))
    cs2_ref, _ = colstats(Pf, "sum")
    torch.cuda.synchronize()
    verdict["config5_slice"] = {
        "colsums_identical": bool(torch.equal(cs, cs_ref)),
        "rowsums_rel_err": float((rs - rs_ref).abs().max()) / max(float(rs_ref.abs().max()), 1e-300),
        "aperm_colsums_identical": bool(torch.equal(cs2, cs2_ref)),
        "shard_nnz": As.nnz, "axis_block": [lo, hi]}
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(verdict, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
