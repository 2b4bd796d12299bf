"""Unary crossprod(x) and SVT x SVT crossprod(x, y) on resident operands: the sparse-aware kernel
(svt_dev_crossprod_csc_csc, kernels_gram.hip) beside the dense-buffer route it replaces
(svt_dev_crossprod_csc_csc_dense_buffer) and, for the unary form, the row-panel `%*%` kernel on (t(x), x).
Shapes: the reference's published ones (inst/scripts/benchmark_crossprod.R:123-166: 25000 x 400 @ 0.07,
25000 x 650 @ 0.20) and BASELINE config-2 scale (1e6 x 1e4 @ 1 % -> 1e4 x 1e4).
usage: sparse_crossprod_time.py [small|big|all] [reps]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import (DeviceCSC, crossprod_csc_csc, crossprod_csc_csc_dense_buffer, matmul_csc_csc, _lib)

what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)


def timed(fn, n=reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def wall(fn, n=1):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def sample_check(A, out, ncheck=40, seed=5):
    """cells of crossprod(A) against a dense product of the two columns (torch, float64)"""
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    n = A.ncol
    worst = 0.0
    for _ in range(ncheck):
        c, j = int(torch.randint(0, n, (1,), generator=g)), int(torch.randint(0, n, (1,), generator=g))
        def col(k):
            b, e = int(A.col_ptr[k]), int(A.col_ptr[k + 1])
            d = torch.zeros(A.nrow, dtype=torch.float64, device=dev)
            d[A.row_idx[b:e].long()] = A.val[b:e].double()
            return d
        dc, dj = col(c), col(j)
        want = float((dc * dj).sum())
        scale = float((dc * dj).abs().sum()) + 1e-300
        got = float(out[j, c])
        worst = max(worst, abs(got - want) / scale)
    return worst


def unary(nrow, ncol, dens, seed, with_dense=True, with_spmm=True, tag=""):
    cp, ri, v = synth.random_device_csc(nrow, ncol, dens, seed=seed, device=dev)
    A = DeviceCSC(nrow, cp, ri, v)
    alg = A.nnz * 12 + (ncol + 1) * 8 + ncol * ncol * 8
    print(f"--- crossprod(x), x {nrow} x {ncol} @ {dens} ({A.nnz} nonzeros){tag}; algorithmic bytes {alg / 1e9:.3f} GB", flush=True)
    ms_t = timed(lambda: A.t())
    At = A.t()
    out = torch.empty((ncol, ncol), dtype=torch.float64, device=dev)
    ws = torch.empty(_lib().svt_dev_crossprod_csc_csc_ws_bytes(At.handle), dtype=torch.uint8, device=dev)
    ms = timed(lambda: crossprod_csc_csc(At, A, sym=True, out=out, ws=ws))
    _, flag = crossprod_csc_csc(At, A, sym=True, out=out, ws=ws)
    torch.cuda.synchronize()
    pairs = float((At.col_ptr[1:] - At.col_ptr[:-1]).double().pow(2).sum()) / 2
    print(f"sparse-aware, symmetric      {ms:9.3f} ms (+ t(x) {ms_t:.3f} ms)  {alg / ms / 1e6:8.1f} GB/s = {alg / ms / 1e6 / 8000:.4f} of 8 TB/s; "
          f"{pairs / ms / 1e6:7.2f} G pairs/s  flag {int(flag.item())}", flush=True)
    sym_ok = bool(torch.equal(out, out.T))
    err = sample_check(A, out)
    print(f"   bit-symmetric {sym_ok}; worst sampled |err| / sum|terms| {err:.2e}", flush=True)
    ms = timed(lambda: crossprod_csc_csc(At, A, sym=False, out=out, ws=ws))
    print(f"sparse-aware, general (x, x) {ms:9.3f} ms", flush=True)
    if with_spmm:
        ws2 = torch.empty(_lib().svt_dev_matmul_csc_csc_ws_bytes(At.handle), dtype=torch.uint8, device=dev)
        out2 = torch.empty((ncol, ncol), dtype=torch.float64, device=dev)
        ms = timed(lambda: matmul_csc_csc(At, A, out=out2, ws=ws2))
        print(f"row-panel %*% kernel on (t(x), x) {ms:9.3f} ms   max |diff| {float((out2 - out).abs().max()):.2e}", flush=True)
        del out2, ws2
    if with_dense:
        out3 = torch.empty((ncol, ncol), dtype=torch.float64, device=dev)
        ms = wall(lambda: crossprod_csc_csc_dense_buffer(A, A, out=out3))
        crossprod_csc_csc(At, A, sym=True, out=out, ws=ws); torch.cuda.synchronize()
        d = float((out3 - out).abs().max())
        print(f"dense-buffer route (rounds 2-5)   {ms:9.3f} ms (wall, allocations inside)   max |diff| {d:.2e}", flush=True)
        del out3
    return A, At


def binary(nrow, nx, dx, ny, dy, seed):
    cp, ri, v = synth.random_device_csc(nrow, nx, dx, seed=seed, device=dev)
    X = DeviceCSC(nrow, cp, ri, v)
    cp, ri, v = synth.random_device_csc(nrow, ny, dy, seed=seed + 1, device=dev)
    Y = DeviceCSC(nrow, cp, ri, v)
    for a, b, name in ((X, Y, "crossprod(svt1, svt2)"), (Y, X, "crossprod(svt2, svt1)")):
        print(f"--- {name}: {nrow} x {a.ncol} @ {a.nnz / nrow / a.ncol:.2f}, {nrow} x {b.ncol} @ {b.nnz / nrow / b.ncol:.2f}", flush=True)
        ms_t = timed(lambda: a.t())
        at = a.t()
        out = torch.empty((b.ncol, a.ncol), dtype=torch.float64, device=dev)
        ws = torch.empty(_lib().svt_dev_crossprod_csc_csc_ws_bytes(at.handle), dtype=torch.uint8, device=dev)
        ms = timed(lambda: crossprod_csc_csc(at, b, out=out, ws=ws))
        print(f"sparse-aware                 {ms:9.3f} ms (+ t(x) {ms_t:.3f} ms)", flush=True)
        out3 = torch.empty_like(out)
        ms = wall(lambda: crossprod_csc_csc_dense_buffer(a, b, out=out3), n=3)
        crossprod_csc_csc(at, b, out=out, ws=ws); torch.cuda.synchronize()
        print(f"dense-buffer route           {ms:9.3f} ms (wall)   max |diff| {float((out3 - out).abs().max()):.2e}", flush=True)


if what in ("small", "all"):
    unary(25000, 400, 0.07, 11, tag=" [reference: svt1]")
    unary(25000, 650, 0.20, 12, tag=" [reference: svt2]")
    binary(25000, 400, 0.07, 650, 0.20, 13)
    unary(100_000, 2000, 0.01, 14)
    unary(100_000, 2000, 0.05, 15)
if what in ("big", "all"):
    unary(1_000_000, 10_000, 0.01, 1, with_dense=(what == "big" or True), with_spmm=True, tag=" [BASELINE config-2 scale]")
