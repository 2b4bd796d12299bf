"""World-size-2 tests of the multi-GPU sharding logic on CPU (gloo).  The local
compute is the CPU oracle here; on the GPU box bench.py plugs the HIP path into
the same functions."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import random_csc
        from oracle import oracle_session
        from sparsearray_amd import SVT_SparseArray
        from sparsearray_amd import parallel as par
        S = oracle_session()
        nrow, ncol, K = 5000, 37, 9
        cp, ri, v = random_csc(nrow, ncol, 0.02, seed=5)
        Y = np.random.default_rng(6).uniform(-1, 1, (nrow, K))
        full = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)

        # crossprod: rows sharded, partials all-reduced
        r0, r1 = par.row_block(nrow, rank, world)
        scp, sri, sv = par.row_shard_csc(cp, ri, v, r0, r1)
        shard = SVT_SparseArray.from_csc((r1 - r0, ncol), "double", scp, sri, sv)
        got = par.sharded_crossprod(
            lambda: torch.from_numpy(np.ascontiguousarray(S.crossprod(shard, Y[r0:r1]))))
        want = S.crossprod(full, Y)
        ok1 = np.allclose(got.numpy(), want, rtol=1e-12, atol=1e-12)

        # colSums: leaves sharded by nnz, scalars gathered
        blocks = par.col_blocks_by_nnz(cp, world)
        c0, c1 = blocks[rank]
        sub = SVT_SparseArray((nrow, c1 - c0), "double", full.leaves[c0:c1])
        loc = torch.from_numpy(np.asarray(S.colSums(sub), dtype=np.float64))
        allc = par.gather_columns(loc, [b[1] - b[0] for b in blocks])
        ok2 = np.array_equal(allc.numpy(), S.colSums(full))
        # torch-tensor flavour of the row filter agrees with the numpy one
        tcp, tri, tv = par.row_shard_csc(torch.from_numpy(cp), torch.from_numpy(ri),
                                         torch.from_numpy(v), r0, r1)
        ok3 = np.array_equal(tcp.numpy(), scp) and np.array_equal(tri.numpy(), sri) \
            and np.array_equal(tv.numpy(), sv)
        # rowsum: leaves sharded by nnz, (ncol_local, ngroup) slabs gathered (ragged blocks)
        grp = list(np.random.default_rng(7).integers(1, 6, nrow))
        loc_rs = np.ascontiguousarray(np.asarray(S.rowsum(sub, grp)[0], dtype=np.float64).T)   # (c1-c0, ngroup)
        allrs = par.sharded_rowsum(None, blocks, None, 5, local=lambda: torch.from_numpy(loc_rs))
        ok4 = np.array_equal(allrs.numpy().T, np.asarray(S.rowsum(full, grp)[0]))
        # colSums of the ROW shards: partial sums all-reduced
        part = par.sharded_colsums_rows(None, local=lambda: torch.from_numpy(
            np.asarray(S.colSums(shard), dtype=np.float64).copy()))
        ok5 = np.allclose(part.numpy(), S.colSums(full), rtol=1e-12, atol=1e-12)
        # slabs of an axis gathered back into place (config 5: results of an array cut along axis 1)
        full3 = torch.arange(5 * 7 * 3, dtype=torch.float64)
        blk = [par.row_block(7, r, world) for r in range(world)]
        lo, hi = blk[rank]
        mine = full3.view(3, 7, 5)[:, lo:hi, :].reshape(-1)
        ok5 = ok5 and bool(torch.equal(par.gather_axis(mine, (5, 7, 3), 1, blk), full3))
        # unary crossprod(A) and crossprod(A, B), both sparse: leaf blocks, all-gather of the other operand
        bcp, bri, bv = par.col_shard_csc(torch.from_numpy(cp), torch.from_numpy(ri), torch.from_numpy(v), c0, c1)

        def prod_with(full_csc):
            fcp, fri, fv = (a.numpy() for a in full_csc)
            other = SVT_SparseArray.from_csc((nrow, len(fcp) - 1), "double", fcp, fri, fv)
            return torch.from_numpy(np.ascontiguousarray(S.crossprod(sub, other)))
        got1 = par.sharded_crossprod_sparse(prod_with, (bcp.contiguous(), bri, bv), blocks)
        ok6 = np.allclose(got1.numpy(), S.crossprod(full), rtol=1e-12, atol=1e-12)
        cpB, riB, vB = random_csc(nrow, 11, 0.05, seed=8)
        fullB = SVT_SparseArray.from_csc((nrow, 11), "double", cpB, riB, vB)
        blocksB = par.col_blocks_by_nnz(cpB, world)
        b0, b1 = blocksB[rank]
        Bblk = par.col_shard_csc(torch.from_numpy(cpB), torch.from_numpy(riB), torch.from_numpy(vB), b0, b1)
        got2 = par.sharded_crossprod_sparse(prod_with, (Bblk[0].contiguous(), Bblk[1], Bblk[2]), blocksB,
                                            blocks_A=blocks)
        ok6 = ok6 and np.allclose(got2.numpy(), S.crossprod(full, fullB), rtol=1e-12, atol=1e-12)
        # colsum: leaves sharded, partials all-reduced (groups of columns span the ranks)
        cgrp = list(np.random.default_rng(9).integers(1, 5, ncol))
        ug = sorted(set(cgrp))
        loc_cs = np.zeros((nrow, len(ug)))
        sub_cs, sub_ug = S.colsum(sub, cgrp[c0:c1])
        for k, gname in enumerate(sub_ug):
            loc_cs[:, ug.index(gname)] = np.asarray(sub_cs)[:, k]
        got_cs = par.sharded_colsum(lambda: torch.from_numpy(loc_cs))
        ok7 = np.allclose(got_cs.numpy(), np.asarray(S.colsum(full, cgrp)[0]), rtol=1e-12, atol=1e-12)
        # 2-D rowSums: owned rows (all-gather) and sharded leaves (all-reduce)
        rblocks = [par.row_block(nrow, r, world) for r in range(world)]
        got_rs = par.sharded_rowsums_2d(lambda: torch.from_numpy(np.asarray(S.rowSums(shard), dtype=np.float64).copy()),
                                        blocks=rblocks)
        got_rs2 = par.sharded_rowsums_2d(lambda: torch.from_numpy(np.asarray(S.rowSums(sub), dtype=np.float64).copy()))
        want_rs = np.asarray(S.rowSums(full))
        ok7 = ok7 and np.array_equal(got_rs.numpy(), want_rs) and np.allclose(got_rs2.numpy(), want_rs, rtol=1e-12, atol=1e-12)
        q.put((rank, ok1, ok2, ok3, ok4, ok5, ok6, ok7))
    finally:
        dist.destroy_process_group()


def test_row_sharded_crossprod_and_column_sharded_colsums_gloo():
    world = 2
    port = 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, ok3, ok4, ok5, ok6, ok7 in res:
        assert ok6, f"rank {rank}: sharded sparse x sparse / unary crossprod differs"
        assert ok7, f"rank {rank}: sharded colsum / 2-D rowSums differ"
        assert ok1, f"rank {rank}: sharded crossprod differs"
        assert ok2, f"rank {rank}: sharded colSums differs"
        assert ok3, f"rank {rank}: torch/numpy row filter differ"
        assert ok4, f"rank {rank}: sharded rowsum differs"
        assert ok5, f"rank {rank}: row-sharded colSums differ"


def _subgroup_worker(rank, world, port, q):
    """Ragged gather inside a sub-group whose member indices are not global ranks
    (advisor finding, round 1: broadcast(src=) takes a global rank)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparsearray_amd import parallel as par
        g = dist.new_group([1, 2])                    # every rank must take part in new_group
        ok = True
        if rank in (1, 2):
            me = rank - 1
            sizes = [3, 5]
            loc = torch.arange(sizes[me], dtype=torch.float64) + 100.0 * rank
            got = par.gather_columns(loc, sizes, group=g)
            want = torch.cat([torch.arange(3, dtype=torch.float64) + 100.0,
                              torch.arange(5, dtype=torch.float64) + 200.0])
            ok = bool(torch.equal(got, want))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_ragged_gather_in_subgroup_gloo():
    world = 3
    port = 31500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _peer_worker(rank, world, port, q):
    """PeerReducer on CPU: the windows are shared memory, the protocol (generation counters, who waits for
    whom, buffer reuse) is the one the GPU path runs."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparsearray_amd import parallel as par
        shape = (3, 50)
        R = par.PeerReducer(shape, torch.float64, "cpu", None, nbuf=2, timeout_s=60)
        base = torch.arange(150, dtype=torch.float64).reshape(shape)
        outs = [torch.zeros(shape, dtype=torch.float64) for _ in range(2)]
        ok = True

        def want(step):
            return (step + 1) * world * (world + 1) / 2 + world * base
        nstep = 9
        for step in range(nstep):
            b = step % 2
            if step > 0:
                R.finish((step - 1) % 2, outs[(step - 1) % 2])
                ok = ok and bool(torch.equal(outs[(step - 1) % 2], want(step - 1)))
            R.before_overwrite(b)
            R.push(b, (rank + 1) * (step + 1) + base)
            if rank == 1 and step == 4:
                import time
                time.sleep(0.3)                    # a rank that falls behind: the others wait for its counters
        R.finish((nstep - 1) % 2, outs[(nstep - 1) % 2])
        ok = ok and bool(torch.equal(outs[(nstep - 1) % 2], want(nstep - 1)))
        # the same sums through the collective
        t = (rank + 1) * nstep + base
        dist.all_reduce(t)
        ok = ok and bool(torch.equal(t, want(nstep - 1)))
        dist.barrier()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_peer_reducer_matches_all_reduce_gloo(world):
    port = 33500 + os.getpid() % 2000 + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_row_blocks_cover():
    sys.path.insert(0, ROOT)
    from sparsearray_amd.parallel import row_block
    for n, w in ((10, 3), (7, 8), (1_000_000, 8)):
        for align in (1, 128):
            blocks = [row_block(n, r, w, align) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            assert all(b[0] % align == 0 or b[0] == n for b in blocks)


def test_blocked_generator_is_rank_count_invariant():
    """The benchmark's global matrix is the same at every N: the union of the row blocks that the
    ranks of an N-way run generate equals the one-rank matrix (sparsearray_amd/synth.py)."""
    sys.path.insert(0, ROOT)
    from sparsearray_amd import synth
    from sparsearray_amd.parallel import row_shard_csc
    cp, ri, v, (r0, r1) = synth.random_device_csc_blocked(20000, 30, 0.01, 3, device="cpu")
    assert (r0, r1) == (0, 20000) and int(cp[-1]) == int(20000 * 30 * 0.01)
    Y = synth.random_dense_blocked(20000, 3, 9, "cpu")
    for world in (2, 4, 8):
        per = 8 // world
        for rank in range(world):
            cpa, ria, va, (a0, a1) = synth.random_device_csc_blocked(
                20000, 30, 0.01, 3, device="cpu", first=rank * per, last=(rank + 1) * per)
            scp, sri, sv = row_shard_csc(cp, ri, v, a0, a1)
            assert torch.equal(scp, cpa) and torch.equal(sri, ria) and torch.equal(sv, va)
            Ya = synth.random_dense_blocked(20000, 3, 9, "cpu", first=rank * per, last=(rank + 1) * per)
            assert torch.equal(Y[:, a0:a1], Ya)


def test_bench_launcher_reports_dead_ranks():
    """`python bench.py --gpus 2` started directly spawns its ranks as a child process (torch.distributed.run) and hands
    their fate on: where the ranks cannot run -- no GPU in this container (and on a one-GPU box rank 1 has no device) --
    the exit status is non-zero and no result line is printed."""
    import subprocess
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs here: the ranks would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--nrow", "4096",
                        "--ncol", "500", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert '"metric"' not in p.stdout
