#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

Workload (BASELINE.json configs[1], reading 2a of SURVEY.md section 8d):
    A = SVT_SparseMatrix 1e6 x 1e4 @ 1% density (randomSparseArray()-style),
    Y = dense double 1e6 x 128,  step = crossprod(A, Y) -> 1e4 x 128.
One step = one pass of the hot path over operands already resident in HBM.
Metric: GNZ/s = nonzeros of A streamed per second (whole job, all ranks).

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak] [--config 2|4]

N > 1 runs one rank per GPU under torch.distributed.run (RCCL): either the caller starts it that way (RANK / WORLD_SIZE
in the environment), or `python bench.py --gpus N` starts its own ranks -- the parent spawns
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a fresh child process before it touches the GPU, relays
rank 0's JSON line as its last line of output and exits with the child's status.  Default `--scaling strong`:
the SAME 1e6 x 1e4 problem at every N -- the matrix is defined as 8 row blocks
(sparsearray_amd/synth.py, random_device_csc_blocked), rank r owns blocks r*8/N .. (r+1)*8/N-1 of A
and of Y (rows = the contracted dimension, sparsearray_amd/parallel.py), and the 10 MB ncol x K
result is all-reduced inside every step, overlapped with the next step's product.
`--scaling weak` gives every rank a full-size block of its own (an N times larger problem).
`--config 4`: BASELINE.json configs[3], 1e7 x 5e4 @ 0.1%, step = crossprod(A, Y 1e7 x 128) + colSums(A),
sharded the same way (all-reduce of the 51 MB product and of the 5e4 column sums).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / hipIpc fail otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    2: dict(nrow=1_000_000, ncol=10_000, density=0.01, K=128,
            name="BASELINE.json configs[1] (reading 2a)"),
    4: dict(nrow=10_000_000, ncol=50_000, density=0.001, K=128,
            name="BASELINE.json configs[3] (crossprod + colSums)"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", type=int, choices=sorted(CONFIGS), default=2)
    p.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    p.add_argument("--nrow", type=int, default=None)
    p.add_argument("--ncol", type=int, default=None)
    p.add_argument("--density", type=float, default=None)
    p.add_argument("--K", type=int, default=None)
    p.add_argument("--path", choices=["pbc", "v1"], default="pbc",
                   help="pbc: panel-blocked LDS kernel (default); v1: gather kernel (1 GPU only)")
    p.add_argument("--cbw", type=int, default=0, help="0 0 0: layout chosen by density (LDS-DMA or gather kernel)")
    p.add_argument("--wpb", type=int, default=0)
    p.add_argument("--logr", type=int, default=0)
    p.add_argument("--backend", default="nccl",
                   help="nccl (= RCCL, the product path); gloo only to rehearse the N > 1 control flow on a one-GPU box")
    p.add_argument("--same-device", action="store_true",
                   help="rehearsal only: every rank uses GPU 0 (needs --backend gloo)")
    p.add_argument("--reduce", choices=["rccl", "peer"], default="rccl",
                   help="N > 1: rccl = all-reduce of the result (default); peer = peer-to-peer copies of the partials + "
                        "a local sum (sparsearray_amd/parallel.py, PeerReducer; no collective kernel)")
    p.add_argument("--spare-cus", type=int, default=-1,
                   help="CUs the product kernel leaves idle (room for RCCL's kernels beside it).  Default: 0 on one GPU; at "
                        "N > 1 with the RCCL reducer the untimed warm-up times a few steps with 0 and with 32 and the timed steps "
                        "run with the faster of the two (the product fills every CU's registers and LDS: whether a collective's "
                        "kernels can run beside it is a property of the node, and leaving CUs idle costs +8 %% product time at an "
                        "eighth of the rows); the line's `multi_gpu` object reports both")
    p.add_argument("--compare-reducers", action="store_true",
                   help="N > 1: after the timed steps, time the same steps with the peer-to-peer reducer as well "
                        "(`multi_gpu.variants`); the RCCL variants (0 and 32 spare CUs) are always there")
    p.add_argument("--event-every", type=int, default=4,
                   help="HIP events around the dominant kernel on every n-th timed step (roofline.kernel_ms = their mean)")
    p.add_argument("--settle-ms", type=float, default=150.0,
                   help="untimed: keep the GPU busy this long with plain streaming reads of Y (torch.sum) before the W warm-up "
                        "steps, so that the W + K steps do not run while the clocks are still ramping up out of idle "
                        "(tools/debug/whole_call_vs_step.py: the first ~20-40 ms after an idle gap run 3-9 %% slower); 0 = off")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--force-collectives", action="store_true",
                   help="create the process group and run every collective even with ONE rank (an all-reduce of one rank is "
                        "the identity): loads RCCL beside libsvt_hip.so and exercises the N > 1 code path on a one-GPU box; "
                        "says nothing about scaling")
    p.add_argument("--no-extras", action="store_true")
    p.add_argument("--no-sparse-crossprod", action="store_true",
                   help="skip extras.sparse_crossprod (unary / SVT x SVT crossprod at the reference's published shapes and at config-2 scale)")
    a = p.parse_args()
    c = CONFIGS[a.config]
    for k in ("nrow", "ncol", "density", "K"):
        if getattr(a, k) is None:
            setattr(a, k, c[k])
    a.spare_auto = a.spare_cus < 0 and a.gpus > 1 and a.reduce == "rccl" and a.path == "pbc"
    if a.spare_cus < 0:
        a.spare_cus = 0
    return a


def multi_gpu_diagnostics(a, dist, par, dev, sc, A, Y, step, finish, kern_ms, ms_per_step, with_colsums):
    """Why the N > 1 line scales the way it does (every rank runs this; collective calls inside): per-rank kernel time,
    the result's all-reduce alone, the product alone, the fraction of the collective hidden behind the next product,
    and the same steps with the other reduction options."""
    world, rank = dist.get_world_size(), dist.get_rank()
    d = {"backend": dist.get_backend(), "world_size": world, "reducer": a.reduce, "spare_cus": a.spare_cus}

    def gathered(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def region(fn, reps):
        """ms per call of fn over reps calls: barrier + synchronize on both sides, MAX over the ranks."""
        fn(); finish(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        finish(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    d["kernel_ms_per_rank"] = gathered(kern_ms)
    buf = torch.zeros_like(sc.outs[0])
    d["allreduce_alone_ms"] = region(lambda: dist.all_reduce(buf), 5)
    d["allreduce_bytes"] = buf.numel() * 8
    lrow = Y.shape[1]
    tmp = torch.zeros_like(sc.outs[0])

    def product_only():
        sc.plan.run(Y, lrow, tmp)
        if with_colsums:
            from sparsearray_amd.device import colstats
            colstats(A, "sum")
    d["product_alone_ms"] = region(product_only, a.steps)
    hidden = d["product_alone_ms"] + d["allreduce_alone_ms"] - ms_per_step
    d["overlap_fraction"] = max(0.0, min(1.0, hidden / d["allreduce_alone_ms"])) if d["allreduce_alone_ms"] > 0 else None
    variants = {}
    if a.reduce == "rccl":
        from sparsearray_amd.device import set_spare_cus
        other = 32 if a.spare_cus == 0 else 0
        set_spare_cus(other)
        try:
            variants[f"rccl_spare_cus_{other}"] = {"ms_per_step": region(step, a.steps)}
        finally:
            set_spare_cus(a.spare_cus)
    if a.compare_reducers and a.reduce != "peer":
        sc2 = par.ShardedCrossprod(A, a.K, None, a.cbw, a.wpb, a.logr, reducer="peer", spare_cus=a.spare_cus)

        def step2():
            sc2.step(Y)
            if with_colsums:
                par.sharded_colsums_rows(A)

        def region2():
            sc2.wait()
        # (the peer reducer has its own wait; `region` calls the main finish, which is idle here)
        step2(); region2(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step2()
        region2(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / a.steps * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        same = bool(torch.allclose(sc2.result(), sc.result(), rtol=1e-11, atol=1e-11))
        variants["peer_copies"] = {"ms_per_step": float(t.item()), "same_result": same}
        sc2.close()
        del sc2
    d["variants"] = variants
    return d


def host_cores() -> int:
    """CPU cores this process may really use: the cgroup quota when there is
    one (a GPU box hands each job a share of a big host), else the affinity."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, int(os.environ.get("SVT_BENCH_CPU_THREADS", "16")))


def cpu_baseline(col_ptr, row_idx, val, Y, nrow, K, nleaves_sample, threads=None):
    """The oracle (CPU restatement of the reference's C/OpenMP path) timed on
    the host cores of this box (`threads` of them; default: all this job may use), on the first
    `nleaves_sample` leaves of A against all K dense columns."""
    import ctypes
    from oracle import load_oracle
    from sparsearray_amd.svt import make_view_from_csc
    lib = load_oracle()
    ns = min(nleaves_sample, col_ptr.numel() - 1)
    cp = col_ptr[: ns + 1].cpu().numpy()
    nz = int(cp[-1])
    ri = row_idx[:nz].cpu().numpy()
    vv = val[:nz].cpu().numpy()
    yh = np.ascontiguousarray(Y.cpu().numpy())          # (K, nrow) == col-major nrow x K
    view = make_view_from_csc((nrow, ns), "double", cp, ri, vv)
    out = np.zeros((K, ns), dtype=np.float64)
    ncores = threads if threads else host_cores()
    lib.orc_set_max_threads(ncores)
    fn = lib.orc_crossprod2_SVT_mat
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    t0 = time.perf_counter()
    rc = fn(ctypes.addressof(view), yh.ctypes.data, nrow, K, 14, 0, out.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0
    return {"value": nz / dt / 1e9, "unit": "GNZ/s", "cores": ncores, "kind": "port",
            "sample": f"first {ns} leaves of A ({nz} nnz) x all {K} dense columns, "
                      f"oracle C/OpenMP path, {dt:.2f} s wall"}, out, ns


# The only operations the reference publishes timings for: unary crossprod(x) and SVT x SVT crossprod(x, y)
# (inst/scripts/benchmark_crossprod.R:123-166; R `system.time()[["user.self"]]`: CPU seconds summed over the OpenMP
# threads, machine and thread count unstated -- context, not a baseline).  Inputs: this repository's generator
# at the script's shapes (rsparsematrix() under set.seed(333) is not reproducible without R).
PUBLISHED_CPU_SECONDS = {"crossprod(svt1)": 0.224, "crossprod(svt1, svt1)": 0.381,
                         "crossprod(svt1, svt2)": 0.641, "crossprod(svt2, svt1)": 0.608}


def sparse_crossprod_extras(dev, A_big, timed):
    """`extras.sparse_crossprod`: the four published cases through the host entry points (svt_crossprod1_SVT,
    svt_crossprod2_SVT_SVT: marshal + PCIe + t(x) + product + result back) and at device level
    (svt_dev_transpose + svt_dev_crossprod_csc_csc), the CPU oracle's wall time beside each at both thread
    settings; and crossprod(A) at BASELINE config-2 scale (A = the bench operand) at device level with GB/s against
    its algorithmic bytes (A once + the ncol x ncol result) next to the dense-buffer route of rounds 2-5."""
    import ctypes
    from oracle import load_oracle
    from sparsearray_amd import _hip as _hipmod, synth
    from sparsearray_amd.device import (DeviceCSC, crossprod_csc_csc, crossprod_csc_csc_dense_buffer, _lib as _dl)
    from sparsearray_amd.svt import make_view_from_csc
    hl, orc = _hipmod.init(), load_oracle()
    for lib, pre in ((hl, "svt_"), (orc, "orc_")):
        getattr(lib, pre + "crossprod1_SVT").argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        getattr(lib, pre + "crossprod2_SVT_SVT").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    ops = {}
    for name, (nr, nc, d, seed) in {"svt1": (25000, 400, 0.07, 11), "svt2": (25000, 650, 0.20, 12)}.items():
        cp, ri, v = synth.random_device_csc(nr, nc, d, seed=seed, device=dev)
        D = DeviceCSC(nr, cp, ri, v)
        h = (cp.cpu().numpy(), ri.cpu().numpy(), v.cpu().numpy())
        ops[name] = (D, make_view_from_csc((nr, nc), "double", *h), h)

    def wall_ms(fn, reps=3):
        fn()
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); best = min(best, (time.perf_counter() - t0) * 1e3)
        return best
    nthr, nthr3 = host_cores(), max(1, host_cores() // 3)
    out = {"published": "inst/scripts/benchmark_crossprod.R:123-166 (CPU seconds over the OpenMP threads; machine and "
                        "thread count unstated); inputs here: own generator at the same shapes and densities",
           "cases": {}}
    for label, xn, yn in (("crossprod(svt1)", "svt1", None), ("crossprod(svt1, svt1)", "svt1", "svt1"),
                          ("crossprod(svt1, svt2)", "svt1", "svt2"), ("crossprod(svt2, svt1)", "svt2", "svt1")):
        X, xv, _ = ops[xn]
        Y, yv, _ = ops[yn] if yn else (None, None, None)
        nx, ny = X.ncol, (Y.ncol if yn else X.ncol)
        hres, ores = np.zeros((ny, nx)), np.zeros((ny, nx))

        def call(lib, pre, res):
            if yn is None:
                rc = getattr(lib, pre + "crossprod1_SVT")(ctypes.addressof(xv), res.ctypes.data)
            else:
                rc = getattr(lib, pre + "crossprod2_SVT_SVT")(ctypes.addressof(xv), ctypes.addressof(yv), res.ctypes.data)
            assert rc == 0
        host_ms = wall_ms(lambda: call(hl, "svt_", hres))
        orc.orc_set_max_threads(nthr)
        cpu_ms = wall_ms(lambda: call(orc, "orc_", ores), 2)
        orc.orc_set_max_threads(nthr3)
        cpu3_ms = wall_ms(lambda: call(orc, "orc_", ores), 2)
        orc.orc_set_max_threads(nthr)
        scale = np.maximum(np.abs(ores), 1e-9)
        o = torch.empty((ny, nx), dtype=torch.float64, device=dev)
        t_ms = timed(lambda: X.t(), 5)
        Xt = X.t()
        ws = torch.empty(_dl().svt_dev_crossprod_csc_csc_ws_bytes(Xt.handle), dtype=torch.uint8, device=dev)
        Yd = X if yn is None or yn == xn else Y
        k_ms = timed(lambda: crossprod_csc_csc(Xt, Yd, sym=(yn is None), out=o, ws=ws), 10)
        out["cases"][label] = {"host_entry_point_ms": host_ms, "device_level_ms": {"t(x)": t_ms, "product": k_ms},
                               "cpu_oracle_wall_ms": {f"{nthr}_threads": cpu_ms, f"{nthr3}_threads_reference_default": cpu3_ms},
                               "published_reference_cpu_seconds": PUBLISHED_CPU_SECONDS[label],
                               "max_rel_err_host_vs_oracle": float(np.max(np.abs(hres - ores) / scale)),
                               "max_rel_err_device_vs_oracle": float(np.max(np.abs(o.cpu().numpy() - ores) / scale))}
        del o, Xt, ws
    # config-2 scale
    A = A_big
    n = A.ncol
    alg = A.nnz * 12 + (n + 1) * 8 + n * n * 8
    t_ms = timed(lambda: A.t(), 3)
    At = A.t()
    o = torch.empty((n, n), dtype=torch.float64, device=dev)
    ws = torch.empty(_dl().svt_dev_crossprod_csc_csc_ws_bytes(At.handle), dtype=torch.uint8, device=dev)
    flag = [None]

    def prod():
        flag[0] = crossprod_csc_csc(At, A, sym=True, out=o, ws=ws)[1]
    k_ms = timed(prod, 5)
    sym_ok = bool(torch.equal(o, o.T))
    # sampled cells against a dense product of the two columns
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    worst = 0.0
    for _ in range(24):
        c, j = (int(x) for x in torch.randint(0, n, (2,), generator=g))
        cols = []
        for k in (c, j):
            b, e = int(A.col_ptr[k]), int(A.col_ptr[k + 1])
            dcol = torch.zeros(A.nrow, dtype=torch.float64, device=dev)
            dcol[A.row_idx[b:e].long()] = A.val[b:e]
            cols.append(dcol)
        want, tot = float((cols[0] * cols[1]).sum()), float((cols[0] * cols[1]).abs().sum()) + 1e-300
        worst = max(worst, abs(float(o[j, c]) - want) / tot)
    del At, ws
    o2 = torch.empty((n, n), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    crossprod_csc_csc_dense_buffer(A, A, out=o2)
    torch.cuda.synchronize(); dense_ms = (time.perf_counter() - t0) * 1e3
    out["crossprod(A)_config2_scale"] = {
        "shape": [A.nrow, n], "nnz": A.nnz, "algorithmic_bytes": alg,
        "device_level_ms": {"t(A)": t_ms, "product_incl_mirror": k_ms},
        "GB/s_product": alg / k_ms / 1e6, "frac_of_8TB/s_product": alg / k_ms / 1e6 / 8000,
        "GB/s_with_t(A)": alg / (k_ms + t_ms) / 1e6,
        "G_pairs_of_nonzeros_per_s": float(((A.nnz / A.nrow) ** 2 / 2 * A.nrow) / k_ms / 1e6),
        "not_finite_flag": int(flag[0].item()), "bit_symmetric": sym_ok,
        "worst_sampled_err_over_sum_abs_terms": worst,
        "dense_buffer_route_ms_rounds_2_to_5": dense_ms,
        "max_abs_diff_vs_dense_buffer_route": float((o - o2).abs().max().item())}
    del o, o2
    return out


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a FRESH child process
    (`python -m torch.distributed.run`; never an exec, and nothing in this process has touched the GPU yet), pass
    the children's output through, print rank 0's JSON line last and return the launcher's exit status (a rank
    that dies makes torch.distributed.run end the others and return non-zero)."""
    import signal
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")          # (torch.distributed.run would set 1 and say so on stderr)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)

    def forward(sig, _frame):
        try:
            os.killpg(child.pid, sig)
        except ProcessLookupError:
            pass
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, forward)
    line_json = None
    for line in child.stdout:
        t = line.strip()
        if t.startswith("{") and '"metric"' in t:
            line_json = t                           # held back: it must be the last line of this process
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    rc = child.wait()
    if line_json is not None:
        print(line_json, flush=True)
    elif rc == 0:
        rc = 1                                      # the ranks ended without a result line
    return rc if rc >= 0 else 128 - rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started {world} rank(s)")
    if a.same_device:
        local = 0
        os.environ["LOCAL_RANK"] = "0"          # (the HIP library binds to LOCAL_RANK)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    coll = world > 1 or a.force_collectives         # are there collectives to run?
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    from sparsearray_amd import parallel as par
    if a.force_collectives:
        par.force_collectives(True)
    from sparsearray_amd import synth
    from sparsearray_amd.device import (CrossprodPlan, DeviceCSC, PbcPlan, colmedians, colstats,
                                        rowsum, rowsums)

    nrow, ncol, K = a.nrow, a.ncol, a.K
    strong = a.scaling == "strong"
    if strong:
        # one global problem, cut into row blocks that do not depend on the number of ranks
        # (the global matrix is DEFINED as 8 independently drawn row blocks, so that N = 1, 2, 4, 8 compute the
        # same product; any other N would be a different problem -- refuse it rather than redefine the matrix)
        if 8 % world != 0:
            raise SystemExit(f"--gpus {world}: strong scaling runs the one {nrow}x{ncol} problem, which is defined "
                             "as 8 row blocks; N must divide 8 (1, 2, 4 or 8).  Use --scaling weak for other N.")
        nblocks = 8
        per = nblocks // world
        col_ptr, row_idx, val, (r0, r1) = synth.random_device_csc_blocked(
            nrow, ncol, a.density, seed=1, device=dev, nblocks=nblocks, first=rank * per, last=(rank + 1) * per)
        Y = synth.random_dense_blocked(nrow, K, seed=101, device=dev, nblocks=nblocks,
                                       first=rank * per, last=(rank + 1) * per)
    else:
        # every rank owns a full-size block of an N times taller matrix
        col_ptr, row_idx, val = synth.random_device_csc(nrow, ncol, a.density, seed=1 + rank, device=dev)
        Y = synth.random_dense(nrow, K, seed=101 + rank, device=dev)
        r0, r1 = rank * nrow, (rank + 1) * nrow
    lrow = r1 - r0
    A = DeviceCSC(lrow, col_ptr, row_idx, val)
    nnz = A.nnz
    with_colsums = a.config == 4
    layout_ms = None
    csum = [None]
    if a.path == "pbc":
        # one-off re-layout of the sparse operand (reported, not part of a step:
        # it depends on A only and is reused by every product with that A)
        # (a small build first: the first launch of every kernel carries the load of the code object,
        # a cost per process, not per operand)
        wcp, wri, wv = synth.random_device_csc(4096, 700, 0.01, seed=99, device=dev)
        del_me = PbcPlan(DeviceCSC(4096, wcp, wri, wv), 64, a.cbw, a.wpb, a.logr)
        del del_me, wcp, wri, wv
        torch.cuda.synchronize()
        t_l = time.perf_counter()
        sc = par.ShardedCrossprod(A, K, None, a.cbw, a.wpb, a.logr, reducer=a.reduce, spare_cus=a.spare_cus)   # (the plan serves any later setting)
        torch.cuda.synchronize()
        layout_ms = (time.perf_counter() - t_l) * 1e3
        auto_gather = (a.cbw, a.wpb, a.logr) == (0, 0, 0) and a.density * 40 * 128 < 12 and lrow >= 4096
        if auto_gather or (a.wpb == 4 and a.logr >= 9):
            # (pbc_auto_layout / pbgx_ok, kernels_mult_pbc.hip: the XCD-paced kernel wants K in whole pairs of 64-wide
            # tiles and 64 row panels)
            g_logr = a.logr if a.logr else (11 if lrow >> 11 >= 64 else 10 if lrow >> 10 >= 64 else 9)
            paced = ((K + 63) // 64) % 2 == 0 and ((lrow + (1 << g_logr) - 1) >> g_logr) >= 64
            kernel_name = "crossprod_pbc_gatherx_kernel" if paced else "crossprod_pbc_gather_kernel"
        else:
            g_logr = 7
            kernel_name = "crossprod_pbc_dma_kernel"

        def step(ev=None):
            sc.step(Y, ev)
            if with_colsums:
                csum[0] = par.sharded_colsums_rows(A)

        def finish():
            sc.wait()

        def result():
            return sc.result()
    else:
        if world > 1:
            raise SystemExit("--path v1 is a one-GPU diagnostic")
        plan = CrossprodPlan(A, K)
        out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
        kernel_name = "crossprod_gather_kernel<double>"

        def step(ev=None):
            plan.prepare(Y, lrow)
            if ev is not None:
                ev[0].record()
            plan.multiply(out, 1, ncol)
            if ev is not None:
                ev[1].record()

        def finish():
            pass

        def result():
            return out

    spare_tuned = None
    if coll and a.spare_auto:
        # part of the untimed warm-up: the same steps with 0 and with 32 CUs left to the collective's kernels; every
        # rank takes the setting with the smaller MAX-over-ranks time
        from sparsearray_amd.device import set_spare_cus
        spare_tuned = {}
        for cand in (0, 32):
            set_spare_cus(cand)
            for _ in range(2):
                step()
            finish(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                step()
            finish(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / 4 * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            spare_tuned[cand] = float(t.item())
        a.spare_cus = min(spare_tuned, key=spare_tuned.get)
        set_spare_cus(a.spare_cus)
    if a.settle_ms > 0:
        # out of idle: generating the inputs and building the layout leaves the GPU mostly waiting for the host
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < a.settle_ms:
            for _ in range(16):
                torch.sum(Y)
            torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    finish()
    torch.cuda.synchronize()
    if coll:
        dist.barrier()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(a.steps)]
    t0 = time.perf_counter()
    # HIP events bracket the dominant kernel on every `--event-every`-th step (default 4; the 4th, 8th, ...: the first
    # timed step starts on an idle GPU and its interval would hold the host's launch latency): a pair of event records
    # costs a step ~13 us of stream bubbles (1.7733 -> 1.7616 ms per step at config 2a with a quarter of them)
    ev_every = max(1, min(a.event_every, a.steps))
    for i in range(a.steps):
        step(evs[i] if i % ev_every == ev_every - 1 else None)
    finish()
    torch.cuda.synchronize()
    if coll:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if coll:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tn = torch.tensor([nnz], dtype=torch.int64, device=dev)
        dist.all_reduce(tn)
        total_nnz = int(tn.item())
    else:
        total_nnz = nnz
    kern_ms = float(np.mean([e0.elapsed_time(e1) for i, (e0, e1) in enumerate(evs) if i % ev_every == ev_every - 1]))
    res_t = result()
    checksum = [float(res_t.sum().item()), float(res_t.abs().sum().item())]
    diag = None
    if coll and a.path == "pbc":
        # (a failure here must not leave the other ranks inside a collective: let it end the process with a
        # non-zero status -- torch.distributed.run then ends the job -- rather than be caught on one rank)
        diag = multi_gpu_diagnostics(a, dist, par, dev, sc, A, Y, step, finish, kern_ms, elapsed / a.steps * 1e3,
                                     with_colsums)

    if rank != 0:
        if coll:
            dist.destroy_process_group()
        return

    # roofline of the dominant kernel (the sparse x dense product) on this rank, per launch
    alg_bytes = nnz * 12 + 8 * (ncol + 1) + lrow * K * 8 + ncol * K * 8
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile) and world == 1:
        try:
            tj = json.load(open(tfile))
            same_load = tj.get("workload_nnz_nominal") == int(nrow * ncol * a.density)
            if same_load and str(tj.get("kernel", "")).startswith(kernel_name.split("<")[0]):
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this " \
                                 "command, corrected as profiles/README.md says: (2 x FETCH_SIZE + WRITE_SIZE) x 1024; not measured in this run)"
        except Exception:
            traffic = None
    if world == 1:
        para = "1 GPU"
    elif strong:
        para = (f"strong scaling: the one {nrow}x{ncol} problem, rows (contracted dimension) of A and Y sharded over "
                f"{world} ranks; " + ("all-reduce of the ncol x K result inside every step, overlapped with the next "
                                      "step's product" if a.reduce == "rccl" else
                                      "partials exchanged by peer-to-peer copies and summed locally (PeerReducer), the sum "
                                      "of a step taken at the start of the next") + " (sparsearray_amd/parallel.py)")
    else:
        para = (f"weak scaling: every rank owns its own {nrow}x{ncol} block of a {world}x taller matrix; all-reduce "
                "of the ncol x K result inside every step")
    what = "crossprod(A, Y)" + (" + colSums(A)" if with_colsums else "")
    res = {
        "metric": "GNZ/s, SVT crossprod(svt, dense) 1e6x1e4 @1% nnz" if a.config == 2 else
                  "GNZ/s, SVT crossprod(svt, dense) + colSums 1e7x5e4 @0.1% nnz",
        "value": total_nnz * a.steps / elapsed / 1e9,
        "unit": "GNZ/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": a.scaling if world > 1 else "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{what}: A[{nrow if strong else nrow * world}x{ncol} SVT @{a.density}], "
                               f"Y[{nrow if strong else nrow * world}x{K} dense f64] -> {ncol}x{K}; "
                               f"{CONFIGS[a.config]['name']}; {a.scaling} scaling",
                   "nnz_total": total_nnz, "nnz_rank0": nnz, "rows_rank0": lrow,
                   "parallelism": para,
                   "result_checksum": {"sum": checksum[0], "abs_sum": checksum[1]}},
        "roofline": {"bound": "hbm", "kernel": kernel_name,
                     "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
                     # priced against HBM as BASELINE.json asks; what the kernel actually waits for (DESIGN.md section 0,
                     # profiles/r04_ceiling_work_corrected.txt, r03_dma_sq_counters.txt) is not HBM:
                     "limiter": ("lds+issue (one LDS read of Y per record x 64 dense columns + scalar/vector issue in strict "
                                 "alternation; HBM traffic is 1.09x algorithmic at 15 % of peak)"
                                 if kernel_name == "crossprod_pbc_dma_kernel" else
                                 "l2->cu gather (every nonzero fetches its row of Yt through the texture-address path)"
                                 if kernel_name.startswith("crossprod_pbc_gather") else "l2 gather"),
                     "fp64_frac": 2.0 * nnz * K / (kern_ms * 1e-3) / 78.6e12,
                     "fp64_peak_TFLOPs": 78.6,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms": kern_ms,
                     "kernel_ms_from": f"HIP events around the kernel on every {max(1, a.event_every)}-th of the timed steps"},
    }
    if diag is not None:
        res["multi_gpu"] = diag
    if layout_ms is not None:
        res["config"]["layout"] = (f"PBC cbw=40 wpb=4 logR={g_logr} (gather kernel)" if kernel_name.startswith("crossprod_pbc_gather")
                                   else "PBC cbw=40 wpb=16 logR=7 (LDS-DMA kernel)") if a.cbw == 0 else \
            f"PBC cbw={a.cbw} wpb={a.wpb} logR={a.logr}"
        res["config"]["layout_build_ms_once_per_operand"] = layout_ms
        if a.settle_ms > 0:
            res["config"]["untimed_settle_ms_before_the_warmup_steps"] = a.settle_ms
        if a.spare_cus:
            res["config"]["spare_cus"] = a.spare_cus
        if spare_tuned is not None:
            res["config"]["spare_cus_chosen_in_warmup_from_ms_per_step"] = {str(k): v for k, v in spare_tuned.items()}
    if world == 1 and not coll and not a.no_extras and a.config == 2:
        def timed(fn, reps=5):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        # the same W + K steps the way rounds 1-4 timed them -- straight out of an idle GPU, no settle phase (ADVICE round 5:
        # the settle phase changed the headline's methodology; both numbers are in every line now)
        torch.cuda.synchronize(); time.sleep(0.5)
        for _ in range(a.warmup):
            step()
        finish(); torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for _ in range(a.steps):
            step()
        finish(); torch.cuda.synchronize()
        ms_idle = (time.perf_counter() - t0_) / a.steps * 1e3
        grp = torch.randint(1, 1001, (lrow,), device=dev, dtype=torch.int32)
        from sparsearray_amd.device import _lib as _devlib
        rs_out = torch.empty(lrow, dtype=torch.float64, device=dev)
        rs_ws = torch.empty(_devlib().svt_dev_rowstats_ws_bytes(lrow, ncol), dtype=torch.uint8, device=dev)
        med_out = torch.empty(ncol, dtype=torch.float64, device=dev)
        med_ws = torch.empty(_devlib().svt_dev_colmedians_ws_bytes(nnz, ncol), dtype=torch.uint8, device=dev)
        ex = {"ms_per_step_out_of_0.5s_of_idle_without_the_settle_phase": ms_idle}
        for name, fn, nbytes in (
            ("colSums", lambda: colstats(A, "sum"), nnz * 8 + ncol * 16),
            ("colVars", lambda: colstats(A, "var1"), nnz * 8 + ncol * 16),
            ("colMedians", lambda: colmedians(A, out=med_out, ws=med_ws), nnz * 8 + ncol * 16),
            ("rowSums", lambda: rowsums(A, out=rs_out, ws=rs_ws), nnz * 12 + lrow * 8),
            ("rowsum_1e3_groups", lambda: rowsum(A, grp, 1000), nnz * 12 + lrow * 4 + 1000 * ncol * 8),
        ):
            ms = timed(fn)
            ex[name] = {"ms": ms, "GNZ/s": nnz / ms / 1e6, "GB/s": nbytes / ms / 1e6}
        from sparsearray_amd.device import RowSumsPlan, RowsumPlan
        # rowsum with the 16-bit group id of every nonzero computed once per (operand, grouping): 10 B/nz streamed
        rwp = RowsumPlan(A, grp, 1000)
        rw_o = rwp.run()
        ex["rowsum_1e3_groups"]["with_group_ids_prepared_once_ms"] = timed(lambda: rwp.run(out=rw_o))
        ex["rowsum_1e3_groups"]["prepare_ms_once_per_operand_and_grouping"] = timed(lambda: RowsumPlan(A, grp, 1000), 3)
        ex["rowsum_1e3_groups"]["same_result"] = bool(torch.allclose(rw_o, rowsum(A, grp, 1000), rtol=1e-12, atol=1e-13))
        del rwp, rw_o
        rsp = RowSumsPlan(A)
        rs_out2 = torch.empty(lrow, dtype=torch.float64, device=dev)
        ms = timed(lambda: rsp.run(out=rs_out2))
        ex["rowSums"]["with_the_run_table_built_once_ms"] = ms
        ex["rowSums"]["same_result"] = bool(torch.allclose(rs_out, rs_out2, rtol=1e-12, atol=1e-13))
        del rsp, rs_out2
        # the same product when the dense operand is not clean / not column-major
        plan0 = sc.plan
        outx = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
        timed(lambda: plan0.run(Y, lrow, outx), 30)          # (out of the idle gap the allocations above left)
        t_clean = timed(lambda: plan0.run(Y, lrow, outx), 20)
        Yp = Y.clone(); Yp[5, lrow // 8 + 1] = float("inf")
        t_inf = timed(lambda: plan0.run(Yp, lrow, outx), 10)
        Yp[7, :] = float("nan")
        t_col = timed(lambda: plan0.run(Yp, lrow, outx), 10)
        del Yp
        Yrm = Y.t().contiguous()
        t_try = timed(lambda: plan0.run(Yrm, K, outx, tr_y=True), 10)
        del Yrm, outx
        ex["crossprod_whole_call"] = {"ms": t_clean, "one_Inf_in_Y_ms": t_inf, "plus_a_NaN_column_ms": t_col,
                                      "Y_given_by_rows_ms": t_try}
        if a.spare_cus == 0:
            # what the multi-GPU option costs on one GPU: 32 CUs (4 per XCD) left to a collective's kernels
            from sparsearray_amd.device import set_spare_cus
            outs = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
            set_spare_cus(32)
            t_sp = timed(lambda: plan0.run(Y, lrow, outs), 10)
            set_spare_cus(0)
            ex["crossprod_whole_call"]["with_32_CUs_left_idle_ms"] = t_sp
            ex["crossprod_whole_call"]["with_32_CUs_left_idle_same_result"] = bool(
                torch.allclose(outs, result(), rtol=1e-11, atol=1e-11))
            del outs
        # what ONE rank of an N-GPU strong-scaling run of this config computes per step (rank 0's row blocks of the same
        # 8-block matrix, no collective), timed like the main loop: 1.77 / this = the scaling the result's all-reduce can
        # only lower
        if strong and (a.nrow, a.ncol) == (CONFIGS[2]["nrow"], CONFIGS[2]["ncol"]):
            share = {}
            for nshare in (2, 4, 8):
                scp, sri, sv, (sr0, sr1) = synth.random_device_csc_blocked(nrow, ncol, a.density, seed=1, device=dev,
                                                                           nblocks=8, first=0, last=8 // nshare)
                sY = synth.random_dense_blocked(nrow, K, seed=101, device=dev, nblocks=8, first=0, last=8 // nshare)
                sA = DeviceCSC(sr1 - sr0, scp, sri, sv)
                ssc = par.ShardedCrossprod(sA, K, None, a.cbw, a.wpb, a.logr)
                for _ in range(5):
                    ssc.step(sY)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for _ in range(50):
                    ssc.step(sY)
                torch.cuda.synchronize()
                share[str(nshare)] = (time.perf_counter() - t0_) / 50 * 1e3
                del ssc, sA, sY, scp, sri, sv
            ex["rank_share_ms_per_step"] = share
            ex["rank_share_speedup_before_the_collective"] = {k: res["ms_per_step"] / v for k, v in share.items()}
        # once-per-operand costs, steady state (second call: code objects loaded, allocator warm)
        def wall(fn, reps=3):
            best = 1e30
            for _ in range(reps):
                torch.cuda.synchronize(); t0_ = time.perf_counter()
                r_ = fn()
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0_) * 1e3)
                del r_
            return best
        ex["once_per_operand"] = {"layout_build_ms": wall(lambda: PbcPlan(A, K, a.cbw, a.wpb, a.logr)),
                                  "transpose_ms": wall(lambda: A.t())}
        # what a FIRST product on a resident CSC operand costs (everything the step excludes): crossprod =
        # layout build + product; A %*% Y = t(A) + layout of t(A) + product
        def first_crossprod():
            pl = PbcPlan(A, K, a.cbw, a.wpb, a.logr)
            o_ = torch.empty((K, ncol), dtype=torch.float64, device=dev)
            pl.run(Y, lrow, o_)
            return o_

        def first_matmul():
            T_ = A.t()
            pl = PbcPlan(T_, K, a.cbw, a.wpb, a.logr)
            o_ = torch.empty((K, lrow), dtype=torch.float64, device=dev)
            pl.run(Y2f, ncol, o_)
            return o_
        Y2f = synth.random_dense(ncol, K, seed=202, device=dev)
        ex["first_call_from_resident_csc"] = {"crossprod_ms": wall(first_crossprod), "matmul_A_Y_ms": wall(first_matmul)}
        del Y2f
        # config 2b: A %*% Y2 (Y2 = ncol x K) = crossprod(t(A), Y2), t(A) and its layout built on device
        T = A.t()
        plan_t = PbcPlan(T, K, a.cbw, a.wpb, a.logr)
        Y2 = synth.random_dense(ncol, K, seed=202, device=dev)
        out2 = torch.empty((K, lrow), dtype=torch.float64, device=dev)
        ms = timed(lambda: plan_t.run(Y2, ncol, out2))
        ex["matmul_A_Y(2b)"] = {"ms": ms, "GNZ/s": nnz / ms / 1e6,
                                "GB/s": (nnz * 12 + ncol * K * 8 + lrow * K * 8) / ms / 1e6}
        del plan_t, T
        # config 3: A %*% B, B = 1e4 x K sparse @ 1 %: the row-panel kernel on A itself (no t(A), no layout, no
        # dense operand; kernels_spmm.hip); bytes as SURVEY.md section 8(d): A once + B + the dense result
        from sparsearray_amd.device import matmul_csc_csc, _lib as _dl
        bcp, bri, bv = synth.random_device_csc(ncol, K, 0.01, seed=303, device=dev)
        Bs = DeviceCSC(ncol, bcp, bri, bv)
        ws3 = torch.empty(_dl().svt_dev_matmul_csc_csc_ws_bytes(A.handle), dtype=torch.uint8, device=dev)
        flag3 = [None]

        def spmm():
            flag3[0] = matmul_csc_csc(A, Bs, out=out2, ws=ws3)[1]
        ms = timed(spmm)
        from sparsearray_amd.device import SpmmPlan
        t_prep = wall(lambda: SpmmPlan(A))
        sp = SpmmPlan(A)
        out2b = torch.empty((K, lrow), dtype=torch.float64, device=dev)
        ms_plan = timed(lambda: sp.run(Bs, out=out2b))
        same_plan = bool(torch.equal(out2, out2b)) or float((out2 - out2b).abs().max().item()) < 1e-12
        del sp, out2b
        Bd = torch.zeros((K, ncol), dtype=torch.float64, device=dev)           # the same product by the dense route
        Bd[torch.repeat_interleave(torch.arange(K, device=dev), bcp[1:] - bcp[:-1]), bri.long()] = bv
        T = A.t()
        plan_t = PbcPlan(T, K, a.cbw, a.wpb, a.logr)
        out3 = torch.empty((K, lrow), dtype=torch.float64, device=dev)
        ms_dense = timed(lambda: plan_t.run(Bd, ncol, out3))
        ex["svt_x_svt2(3)"] = {"ms": ms, "GNZ/s": nnz / ms / 1e6,
                               "GB/s": (nnz * 12 + Bs.nnz * 12 + lrow * K * 8) / ms / 1e6,
                               "with_A_prepared_once_ms": ms_plan, "prepare_A_ms_once_per_operand": t_prep,
                               "prepared_same_result": same_plan,
                               "nnz_B": int(Bs.nnz), "not_finite_flag": int(flag3[0].item()),
                               "dense_route_ms": ms_dense,
                               "max_abs_diff_vs_dense_route": float((out2 - out3).abs().max().item())}
        del plan_t, T, out2, out3, Bd, Bs, ws3
        if not a.no_sparse_crossprod:
            ex["sparse_crossprod"] = sparse_crossprod_extras(dev, A, timed)
        res["extras"] = ex
        # row f2 of SURVEY.md section 8: the .Call-shaped entry point on HOST leaves -- marshal (src/SVT_SparseArray_class.c:598-633
        # walks the tree the same way) + PCIe both ways + layout build + product, and the same call with the operand
        # kept resident on the device between calls (svt_resident_set_limit)
        import ctypes
        from sparsearray_amd import _hip as _hipmod
        from sparsearray_amd.svt import make_view_from_csc
        hl = _hipmod.init()
        hcp, hri, hv = col_ptr.cpu().numpy(), row_idx.cpu().numpy(), val.cpu().numpy()
        hview = make_view_from_csc((lrow, ncol), "double", hcp, hri, hv)
        hy = np.ascontiguousarray(Y.cpu().numpy())            # (K, nrow) C-order = column-major nrow x K
        hout = np.zeros((K, ncol))
        hfn = hl.svt_crossprod2_SVT_mat
        hfn.restype = ctypes.c_int
        hfn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        hl.svt_resident_set_limit.argtypes = [ctypes.c_size_t]

        def host_call():
            t0_ = time.perf_counter()
            rc_ = hfn(ctypes.addressof(hview), hy.ctypes.data, lrow, K, 14, 0, hout.ctypes.data)
            assert rc_ == 0
            return (time.perf_counter() - t0_) * 1e3
        host_call()
        cold = min(host_call() for _ in range(2))
        hl.svt_resident_set_limit(8 << 30)
        host_call()
        resident = min(host_call() for _ in range(3))
        hl.svt_resident_set_limit(0)
        hl.svt_resident_clear()
        herr = float(np.max(np.abs(hout - result().cpu().numpy()) / np.maximum(np.abs(hout), 1e-12)))
        ex["host_entry_point_ms"] = {"svt_crossprod2_SVT_mat_cold": cold, "with_the_resident_cache": resident,
                                     "bytes_over_pcie_cold": int(nnz * 12 + lrow * K * 8 + ncol * K * 8),
                                     "GNZ/s_cold": nnz / cold / 1e6, "GNZ/s_resident": nnz / resident / 1e6,
                                     "max_rel_diff_vs_device_level_result": herr}
        if not a.no_sparse_crossprod and "sparse_crossprod" in ex:
            # the unary crossprod(A) of the same operand through its .Call-shaped entry point: marshal + 1.2 GB up + t(A) + the
            # sparse-aware kernel + the 0.8 GB result back (the reference's loop nest needs ~5e11 multiply-adds for it)
            h1 = hl.svt_crossprod1_SVT
            h1.restype = ctypes.c_int
            h1.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            hsq = np.zeros((ncol, ncol))

            def host_call1():
                t0_ = time.perf_counter()
                assert h1(ctypes.addressof(hview), hsq.ctypes.data) == 0
                return (time.perf_counter() - t0_) * 1e3
            host_call1()
            ex["sparse_crossprod"]["crossprod(A)_config2_scale"]["host_entry_point_svt_crossprod1_SVT_ms"] = min(host_call1() for _ in range(2))
            ex["sparse_crossprod"]["crossprod(A)_config2_scale"]["host_result_bit_symmetric"] = bool(np.array_equal(hsq, hsq.T))
            del hsq
        del hcp, hri, hv, hy, hout, hview
    if world == 1 and not a.no_cpu_baseline:
        ns = max(1, min(ncol, int(1e8 / max(nnz / ncol, 1))))   # <= 1e8 nz x K: 10-30 core-seconds
        cb, ref_out, ns = cpu_baseline(col_ptr, row_idx, val, Y, lrow, K, ns)
        # the timed GPU result must agree with the CPU oracle on the sample
        got = result()[:, :ns].cpu().numpy()
        err = np.max(np.abs(got - ref_out) / np.maximum(np.abs(ref_out), 1e-12))
        cb["max_rel_err_vs_gpu"] = float(err)
        # the reference's DEFAULT team: min(omp_get_max_threads(), nprocs %/% 3), at least 1 (R/thread-control.R:46-57);
        # a third of the sample so that both legs together stay within ~30 core-seconds
        nthr3 = max(1, host_cores() // 3)
        cb3, _, ns3 = cpu_baseline(col_ptr, row_idx, val, Y, lrow, K, max(1, ns // 3), threads=nthr3)
        cb["at_reference_default_threads"] = {"value": cb3["value"], "unit": "GNZ/s", "cores": nthr3,
                                              "rule": "min(OMP max threads, nprocs %/% 3), R/thread-control.R:46-57",
                                              "sample": cb3["sample"]}
        res["cpu_baseline"] = cb
    print(json.dumps(res))
    if coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
