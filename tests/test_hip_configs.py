"""BASELINE.json configs 3 and 5 at full size on one MI355X, and the two-rank rehearsal of the
multi-GPU path on one device.  The oracle cannot finish these sizes in seconds: parity is checked
against plain torch statements of the same operation (gathers, scatter-adds, sorts of linear
indices) and through identities; integer/index results bit for bit.
Reference: src/SparseMatrix_mult.c:1037-1101 (A %*% B via crossprod2 on t(A)),
src/rowsum_methods.c:281-325, src/SparseArray_aperm.c:892-970."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NROW, NCOL, K = 1_000_000, 10_000, 128


@pytest.fixture(scope="module")
def config3(hip):
    from sparsearray_amd import synth
    from sparsearray_amd.device import DeviceCSC
    dev = torch.device("cuda", 0)
    cp, ri, v = synth.random_device_csc(NROW, NCOL, 0.01, seed=3, device=dev)
    bcp, bri, bv = synth.random_device_csc(NCOL, K, 0.01, seed=33, device=dev)
    return DeviceCSC(NROW, cp, ri, v), DeviceCSC(NCOL, bcp, bri, bv)


def _dense_of(B):
    """(K, nrow) C-contiguous = column-major nrow x K dense copy of a sparse operand (torch)."""
    d = torch.zeros((B.ncol, B.nrow), dtype=torch.float64, device=B.val.device)
    cols = torch.repeat_interleave(torch.arange(B.ncol, device=B.val.device), B.col_ptr[1:] - B.col_ptr[:-1])
    d[cols, B.row_idx.long()] = B.val
    return d


def test_config3_matmul_sparse_sparse_full_size(config3):
    """A %*% B = crossprod(t(A), B): device transposition, panel-blocked layout of t(A), product
    kernel against the densified B.  Sampled result rows are recomputed with torch gathers from A
    itself (not from the transposed copy); B = ones turns every result column into rowSums(A)."""
    from sparsearray_amd.device import PbcPlan, rowsums
    A, B = config3
    At = A.t()
    assert At.nrow == NCOL and At.ncol == NROW and At.nnz == A.nnz
    plan = PbcPlan(At, K)
    Bd = _dense_of(B)                                          # (K, ncol)
    out = torch.empty((K, NROW), dtype=torch.float64, device=Bd.device)    # column-major nrow x K
    plan.run(Bd, NCOL, out)
    torch.cuda.synchronize()
    leaf_of = torch.repeat_interleave(torch.arange(NCOL, device=Bd.device), A.col_ptr[1:] - A.col_ptr[:-1])
    rows = torch.randint(0, NROW, (24,), generator=torch.Generator().manual_seed(3)).tolist() + [0, NROW - 1]
    worst = 0.0
    for r in rows:
        hit = (A.row_idx == r).nonzero().flatten()
        cols, vals = leaf_of[hit], A.val[hit]
        prod = Bd[:, cols] * vals                               # (K, nz in the row)
        want = prod.sum(dim=1)
        scale = prod.abs().sum(dim=1).clamp_min(1e-300)
        worst = max(worst, float(((out[:, r] - want).abs() / scale).max()))
    assert worst <= 1e-12, worst
    del leaf_of
    ones = torch.ones_like(Bd)
    plan.run(ones, NCOL, out)
    rs = rowsums(A)
    torch.cuda.synchronize()
    assert float((out - rs[None, :]).abs().max()) <= 1e-12 * max(1.0, float(rs.abs().max()))


def _sampled_rows_err(out, A, Bd, rows):
    """max over the sampled result rows of |out[:, r] - sum_j A[r, j] * B[j, :]| / sum |terms|, the sum
    recomputed from A itself with torch gathers."""
    leaf_of = torch.repeat_interleave(torch.arange(A.ncol, device=Bd.device), A.col_ptr[1:] - A.col_ptr[:-1])
    worst = 0.0
    for r in rows:
        hit = (A.row_idx == r).nonzero().flatten()
        prod = Bd[:, leaf_of[hit]] * A.val[hit].double()       # (K, nz in the row)
        scale = prod.abs().sum(dim=1).clamp_min(1e-300)
        worst = max(worst, float(((out[:, r] - prod.sum(dim=1)).abs() / scale).max()))
    return worst


def test_config3_matmul_row_panel_kernel_full_size(hip, config3):
    """BASELINE config 3 through the kernel `svt %*% svt2` actually takes (spmm_csc_csc_kernel: row panels of A
    itself, no t(A), no dense operand), per call (svt_dev_matmul_csc_csc) and with the operand-only work done
    once (SpmmPlan): 26 result rows recomputed with torch from A and B; the whole result against the dense
    route (t(A) + panel-blocked layout + densified B); columns of B that partition the inner dimension with
    ones -> the result's columns add up to rowSums(A); an integer pair bit for bit; one Inf in B raises the
    flag and the entry point (svt_matmul_SVT_SVT) returns what the dense route does.
    Reference: src/SparseMatrix_mult.c:728-820, 1037-1101."""
    from sparsearray_amd import SVT_SparseArray
    from sparsearray_amd.device import DeviceCSC, PbcPlan, SpmmPlan, matmul_csc_csc, rowsums
    from helpers import assert_equal
    A, B = config3
    dev = A.val.device
    rows = torch.randint(0, NROW, (24,), generator=torch.Generator().manual_seed(13)).tolist() + [0, NROW - 1]
    Bd = _dense_of(B)
    out1, flag1 = matmul_csc_csc(A, B)
    plan = SpmmPlan(A)
    out2, flag2 = plan.run(B)
    torch.cuda.synchronize()
    assert int(flag1.item()) == 0 and int(flag2.item()) == 0
    assert _sampled_rows_err(out1, A, Bd, rows) <= 1e-12
    assert _sampled_rows_err(out2, A, Bd, rows) <= 1e-12
    # the dense route on the same operands
    At = A.t()
    dense_plan = PbcPlan(At, K)
    ref = torch.empty((K, NROW), dtype=torch.float64, device=dev)
    dense_plan.run(Bd, NCOL, ref)
    torch.cuda.synchronize()
    top = float(ref.abs().max())
    assert float((out1 - ref).abs().max()) <= 1e-12 * top
    assert float((out2 - ref).abs().max()) <= 1e-12 * top
    del out2
    # ones at rows j with j % K == k in column k: the K result columns add up to rowSums(A)
    j = torch.arange(NCOL, device=dev)
    order = torch.argsort(j % K, stable=True)
    cp1 = torch.zeros(K + 1, dtype=torch.int64, device=dev)
    cp1[1:] = torch.cumsum(torch.bincount(j % K, minlength=K), 0)
    B1 = DeviceCSC(NCOL, cp1, order.to(torch.int32), torch.ones(NCOL, dtype=torch.float64, device=dev))
    out3, flag3 = plan.run(B1)
    rs = rowsums(A)
    torch.cuda.synchronize()
    assert int(flag3.item()) == 0
    assert float((out3.sum(dim=0) - rs).abs().max()) <= 1e-12 * max(1.0, float(rs.abs().max()))
    del out3, B1
    # integer operands: every product and partial sum is an integer far below 2^53 -> any order of
    # additions gives the same bits as the dense route on the same values held as doubles
    vi = (A.val * 100).round().to(torch.int32); vi[vi == 0] = 7
    bi = (B.val * 100).round().to(torch.int32); bi[bi == 0] = -3
    Ai, Bi = DeviceCSC(NROW, A.col_ptr, A.row_idx, vi), DeviceCSC(NCOL, B.col_ptr, B.row_idx, bi)
    outi, flagi = matmul_csc_csc(Ai, Bi)
    Ati = DeviceCSC(NCOL, At.col_ptr, At.row_idx, (At.val * 100).round())
    Ati.val[Ati.val == 0] = 7.0
    Bid = _dense_of(DeviceCSC(NCOL, B.col_ptr, B.row_idx, bi.double()))
    PbcPlan(Ati, K).run(Bid, NCOL, ref)
    torch.cuda.synchronize()
    assert int(flagi.item()) == 0
    assert torch.equal(outi, ref)
    assert _sampled_rows_err(outi, Ai, Bid, rows[:6]) == 0.0
    del outi, Ai, Bi, Ati, Bid, vi, bi
    # one Inf in B: the flag goes up (the result is void) ...
    bv = B.val.clone(); bv[5] = float("inf")
    Binf = DeviceCSC(NCOL, B.col_ptr, B.row_idx, bv)
    _, flag4 = matmul_csc_csc(A, Binf)
    _, flag5 = plan.run(Binf)
    torch.cuda.synchronize()
    assert int(flag4.item()) != 0 and int(flag5.item()) != 0
    # ... and the entry point returns the dense route's result (the reference's dirty-leaf loops multiply the
    # implicit zeros of A too: a whole result column of NaN, src/SparseVec_dotprod.c:48-65)
    dense_plan.run(_dense_of(Binf), NCOL, ref)
    torch.cuda.synchronize()
    x = SVT_SparseArray.from_csc((NROW, NCOL), "double", A.col_ptr.cpu().numpy(), A.row_idx.cpu().numpy(),
                                 A.val.cpu().numpy())
    y = SVT_SparseArray.from_csc((NCOL, K), "double", B.col_ptr.cpu().numpy(), B.row_idx.cpu().numpy(),
                                 bv.cpu().numpy())
    got = np.asarray(hip.matmul(x, y))
    assert got.shape == (NROW, K)
    assert_equal(got, ref.cpu().numpy().T, tol=1e-12, atol=1e-13 * top, what="x %*% y with one Inf in y")
    assert np.isnan(got).any()


def test_config3_rowsum_1e3_groups_full_size(config3):
    """rowsum(A, group) with 1e3 groups on 1e6 x 1e4 against torch.index_add_ on the flattened
    ngroup x ncol result (src/rowsum_methods.c:44-64, 281-325)."""
    from sparsearray_amd.device import rowsum
    A, _ = config3
    dev = A.val.device
    ng = 1000
    g = torch.Generator(device=dev); g.manual_seed(9)
    grp = torch.randint(1, ng + 1, (NROW,), generator=g, device=dev, dtype=torch.int32)
    got = rowsum(A, grp, ng)                                   # (ncol, ngroup) = column-major ngroup x ncol
    leaf_of = torch.repeat_interleave(torch.arange(NCOL, device=dev), A.col_ptr[1:] - A.col_ptr[:-1])
    cell = leaf_of * ng + (grp[A.row_idx.long()].long() - 1)
    del leaf_of
    want = torch.zeros(NCOL * ng, dtype=torch.float64, device=dev)
    want.index_add_(0, cell, A.val)
    absum = torch.zeros(NCOL * ng, dtype=torch.float64, device=dev)
    absum.index_add_(0, cell, A.val.abs())
    torch.cuda.synchronize()
    err = (got.flatten() - want).abs() / absum.clamp_min(1e-300)
    assert float(err.max()) <= 1e-12
    assert float(got.abs().sum()) > 0


# ---------------------------------------------------------------------------
# config 5: aperm of the 2e4 x 2e4 x 64 array
# ---------------------------------------------------------------------------
D5 = (20_000, 20_000, 64)


@pytest.mark.parametrize("perm", [(1, 3, 2), (3, 1, 2), (2, 1, 3), (2, 3, 1), (3, 2, 1)])
def test_config5_aperm_full_size(hip, perm):
    """aperm(x, perm) on 1.28e8 nonzeros against a torch sort of the permuted linear indices:
    col_ptr, offsets and values bit for bit (src/SparseArray_aperm.c:892-970)."""
    from sparsearray_amd import synth
    from sparsearray_amd.device import DeviceCSC
    dev = torch.device("cuda", 0)
    cp, ri, v = synth.random_device_csc(D5[0], D5[1] * D5[2], 0.005, seed=5, device=dev)
    A = DeviceCSC(D5[0], cp, ri, v)
    P, new_dim = A.aperm(D5, perm)
    torch.cuda.synchronize()
    assert new_dim == tuple(D5[p - 1] for p in perm)
    # subscripts of every nonzero in the original array
    leaf = torch.repeat_interleave(torch.arange(A.ncol, device=dev), cp[1:] - cp[:-1])
    sub = [ri.long(), leaf % D5[1], leaf // D5[1]]
    del leaf
    nsub = [sub[p - 1] for p in perm]
    lin = nsub[0] + new_dim[0] * (nsub[1] + new_dim[1] * nsub[2])
    del sub, nsub
    order = torch.argsort(lin)
    lin = lin[order]
    want_rows = (lin % new_dim[0]).to(torch.int32)
    want_leaf = lin // new_dim[0]
    del lin
    assert torch.equal(P.row_idx, want_rows)
    del want_rows
    assert torch.equal(P.val, v[order])
    del order
    want_cp = torch.zeros(new_dim[1] * new_dim[2] + 1, dtype=torch.int64, device=dev)
    want_cp[1:] = torch.cumsum(torch.bincount(want_leaf, minlength=new_dim[1] * new_dim[2]), 0)
    assert torch.equal(P.col_ptr, want_cp)


def test_aperm_more_than_2e31_new_leaves(hip):
    """aperm(x, c(3, 1, 2)) of a 50000 x 50000 x 2 array: 2.5e9 new leaves (20 GB of leaf pointers) -- past what the
    composed route and the 32-bit keys take: 64-bit (new linear index) keys through the library's own radix sort
    (svt_sort.h, five passes of 8 bits; rounds 1-4: rocprim).  Against a torch sort; src/SparseArray_aperm.c:892-970."""
    from sparsearray_amd import synth
    from sparsearray_amd.device import DeviceCSC
    dev = torch.device("cuda", 0)
    D = (50_000, 50_000, 2)
    perm = (3, 1, 2)
    cp, ri, v = synth.random_device_csc(D[0], D[1] * D[2], 2e-4, seed=55, device=dev)          # 1e6 nonzeros
    A = DeviceCSC(D[0], cp, ri, v)
    P, new_dim = A.aperm(D, perm)
    torch.cuda.synchronize()
    assert new_dim == (2, 50_000, 50_000)
    leaf = torch.repeat_interleave(torch.arange(A.ncol, device=dev), cp[1:] - cp[:-1])
    sub = [ri.long(), leaf % D[1], leaf // D[1]]
    nsub = [sub[p - 1] for p in perm]
    lin = nsub[0] + new_dim[0] * (nsub[1] + new_dim[1] * nsub[2])
    order = torch.argsort(lin)
    lin = lin[order]
    assert torch.equal(P.row_idx, (lin % new_dim[0]).to(torch.int32))
    assert torch.equal(P.val, v[order])
    want_leaf = lin // new_dim[0]
    # leaf pointers: 2.5e9 + 1 of them; compared through the positions where they step (all others repeat their neighbour)
    nl = new_dim[1] * new_dim[2]
    assert P.col_ptr.numel() == nl + 1 and int(P.col_ptr[0]) == 0 and int(P.col_ptr[-1]) == A.nnz
    uq, first = torch.unique_consecutive(want_leaf, return_inverse=False, return_counts=True)
    starts = torch.cumsum(first, 0) - first
    assert torch.equal(P.col_ptr[uq], starts) and torch.equal(P.col_ptr[uq + 1], starts + first)
    d = P.col_ptr[1:] - P.col_ptr[:-1]
    assert int(d.min()) >= 0 and int((d != 0).sum()) == uq.numel()


# ---------------------------------------------------------------------------
# two ranks, one device: the HIP path under the sharding + collectives of parallel.py
# ---------------------------------------------------------------------------
def test_two_ranks_same_device_sharded_ops_match_one_rank(hip, tmp_path):
    """Launches tests/workers/dist_gpu_worker.py as a 2-rank torch.distributed.run job (gloo, both
    ranks on GPU 0) and checks its verdict: row-sharded crossprod + all-reduce, row-sharded
    colSums + all-reduce, leaf-sharded colVars and rowsum + gather, on slices of configs 2a and 4, and
    a slice of config 5 cut along axis 1 (per-leaf column sums, rowSums over axis 2, aperm of the shard),
    each against the one-rank result."""
    torch.cuda.empty_cache()
    out = tmp_path / "verdict.json"
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    port = 29000 + os.getpid() % 1500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "workers", "dist_gpu_worker.py"), str(out)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    verdict = json.loads(out.read_text())
    assert set(verdict) == {"config2a_slice", "config4_slice", "config5_slice"}
    v5 = verdict.pop("config5_slice")
    assert v5["colsums_identical"] and v5["aperm_colsums_identical"], v5
    assert v5["rowsums_rel_err"] <= 1e-12, v5
    for name, v in verdict.items():
        assert v["same_shard"], name
        assert v["crossprod_rel_err"] <= 1e-12, (name, v)
        assert v["crossprod_peer_rel_err"] <= 1e-12, (name, v)        # PeerReducer: peer copies + local sum
        assert v["colsums_rel_err"] <= 1e-12, (name, v)
        assert v["sparse_crossprod_rel_err"] <= 1e-12, (name, v)  # leaf-sharded unary crossprod, sparse-aware kernel
        assert v["colvars_identical"], name
        assert v["rowsum_rel_err"] <= 1e-12, (name, v)


def _bench_line(args, nproc, port):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
               "--gpus", str(nproc), "--backend", "gloo", "--same-device"] + args
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_bench_strong_scaling_rehearsal_same_problem(hip):
    """bench.py --gpus 2 (strong scaling, two ranks rehearsed on one device over gloo) computes the SAME
    product as --gpus 1: the global matrix is defined by row blocks that do not depend on the number of
    ranks, the all-reduced result has the same checksum, and the line names the workload and the scaling."""
    torch.cuda.empty_cache()
    common = ["--nrow", "262144", "--ncol", "4000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    one = _bench_line(common, 1, 0)
    two = _bench_line(common + ["--spare-cus", "32", "--compare-reducers"], 2, 29700 + os.getpid() % 200)
    assert two["config"]["spare_cus"] == 32                   # CUs left to the collective
    # the N > 1 line says why it scales the way it does
    m = two["multi_gpu"]
    assert m["backend"] == "gloo" and m["world_size"] == 2 and m["reducer"] == "rccl" and m["spare_cus"] == 32
    assert len(m["kernel_ms_per_rank"]) == 2 and min(m["kernel_ms_per_rank"]) > 0
    assert m["allreduce_alone_ms"] > 0 and m["product_alone_ms"] > 0 and 0.0 <= m["overlap_fraction"] <= 1.0
    assert m["allreduce_bytes"] == 4000 * 128 * 8
    assert m["variants"]["rccl_spare_cus_0"]["ms_per_step"] > 0
    assert m["variants"]["peer_copies"]["ms_per_step"] > 0 and m["variants"]["peer_copies"]["same_result"]
    peer = _bench_line(common + ["--reduce", "peer"], 2, 29950 + os.getpid() % 200)     # PeerReducer instead of all-reduce
    auto = _bench_line(common, 2, 30200 + os.getpid() % 200)          # spare CUs chosen in the warm-up
    tuned = auto["config"]["spare_cus_chosen_in_warmup_from_ms_per_step"]
    assert set(tuned) == {"0", "32"} and auto["multi_gpu"]["spare_cus"] == int(min(tuned, key=tuned.get))
    assert abs(auto["config"]["result_checksum"]["abs_sum"] - one["config"]["result_checksum"]["abs_sum"]) \
        <= 1e-12 * one["config"]["result_checksum"]["abs_sum"]
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and peer["n_gpus"] == 2
    c = peer["config"]["result_checksum"]
    assert abs(c["abs_sum"] - one["config"]["result_checksum"]["abs_sum"]) <= 1e-12 * abs(c["abs_sum"])
    assert "PeerReducer" in peer["config"]["parallelism"]
    assert two["scaling"] == "strong" and "strong scaling" in two["config"]["workload"]
    assert one["config"]["nnz_total"] == two["config"]["nnz_total"]
    a, b = one["config"]["result_checksum"], two["config"]["result_checksum"]
    assert abs(a["abs_sum"] - b["abs_sum"]) <= 1e-12 * abs(a["abs_sum"])
    assert abs(a["sum"] - b["sum"]) <= 1e-9 * abs(a["abs_sum"])
    for line in (one, two):
        for key in ("metric", "value", "unit", "ms_per_step", "roofline", "dtype", "data", "config"):
            assert key in line
        assert line["roofline"]["bound"] == "hbm" and line["roofline"]["frac"] > 0


def test_bench_starts_its_own_ranks(hip):
    """`python3 bench.py --gpus 2 ...` started DIRECTLY (the shape of the driver's one-GPU command, no launcher, no
    WORLD_SIZE in the environment): the parent spawns torch.distributed.run as a fresh child process before it touches
    the GPU, the last line of its output is rank 0's JSON line, and a rank that dies gives a non-zero exit status
    (the reference is single-process, R/thread-control.R:87-92)."""
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--nrow", "262144", "--ncol", "4000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device"] + common
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["multi_gpu"]["world_size"] == 2 and line["multi_gpu"]["backend"] == "gloo"
    assert line["value"] > 0 and line["scaling"] == "strong"
    # a rank that dies: without --same-device rank 1 asks for GPU 1 of a one-GPU box (on a larger node: a rank count
    # that strong scaling refuses on every rank)
    bad = ["--gpus", "2", "--backend", "gloo"] if torch.cuda.device_count() < 2 else ["--gpus", "3", "--backend", "gloo", "--same-device"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + bad + common, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode != 0
    assert not p.stdout.strip().splitlines() or '"metric"' not in p.stdout.strip().splitlines()[-1]
