"""Differential fuzzing of the column-statistics launch paths (thread / 16-lane / wavefront / workgroup
per generalized column, split long segments) and the row-sum panels against plain torch reductions.
Run on the GPU box:  python tools/debug/fuzz_stats.py [ncases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd.device import DeviceCSC, colstats, rowsums

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
shapes = [  # (nrow, nleaves, density, inner)
    (8, 20000, 0.2, 1), (8, 20000, 0.2, 4),           # thread per column
    (3000, 800, 0.03, 1), (3000, 800, 0.03, 8),        # 16-lane groups
    (20000, 300, 0.02, 1),                              # wavefront per column
    (100000, 64, 0.05, 1), (100000, 64, 0.05, 4),      # workgroup per column
    (400000, 6, 0.5, 1), (200000, 40, 0.4, 20),        # split long segments
    (1, 5000, 0.5, 1), (257, 1, 0.9, 1), (1000, 10, 0.0, 1),
    (50000, 5000, 0.004, 1), (33000, 2400, 0.01, 3),    # row sums: 8192-row panels, strata ranges
    (16384, 900, 0.05, 1), (16383, 900, 0.05, 1),       # on either side of the long-panel threshold
]
worst = 0.0
for case in range(ncases):
    nrow, ncol, dens, inner = shapes[case % len(shapes)]
    dens = dens * float(rng.choice([0.5, 1.0, 1.5]))
    counts = rng.binomial(nrow, min(dens, 1.0), size=ncol)
    if case % 5 == 0 and ncol > 3:
        counts[rng.integers(0, ncol, max(1, ncol // 10))] = 0            # empty leaves
    cp = np.zeros(ncol + 1, dtype=np.int64); cp[1:] = np.cumsum(counts)
    nnz = int(cp[-1])
    ri = np.concatenate([np.sort(rng.choice(nrow, size=c, replace=False)) for c in counts]).astype(np.int32) if nnz else np.zeros(0, np.int32)
    v = np.round(rng.normal(size=nnz), 3); v[v == 0] = 0.25
    A = DeviceCSC.from_host(nrow, cp, ri, v)
    nseg = ncol // inner
    vt = torch.as_tensor(v, device=dev)
    seg_of_leaf = torch.arange(ncol, device=dev) // inner
    seg = torch.repeat_interleave(seg_of_leaf, torch.as_tensor(counts, device=dev))
    seg_len = float(nrow * inner)
    s_ref = torch.zeros(nseg, dtype=torch.float64, device=dev).index_add_(0, seg, vt)
    got, _ = colstats(A, "sum", inner=inner)
    err = float((got - s_ref).abs().max() / (1.0 + s_ref.abs().max())) if nseg else 0.0
    mean = s_ref / seg_len
    d = vt - mean[seg]
    ss = torch.zeros(nseg, dtype=torch.float64, device=dev).index_add_(0, seg, d * d)
    nzs = torch.zeros(nseg, dtype=torch.float64, device=dev).index_add_(0, seg, torch.ones_like(vt))
    var_ref = (ss + mean * mean * (seg_len - nzs)) / (seg_len - 1.0) if seg_len > 1 else None
    gv, _ = colstats(A, "var1", inner=inner)
    if var_ref is not None and nseg:
        err = max(err, float((gv - var_ref).abs().max() / (1e-12 + var_ref.abs().max())))
    mn_ref = torch.full((nseg,), float("inf"), dtype=torch.float64, device=dev).scatter_reduce_(0, seg, vt, "amin")
    mn_ref = torch.where(nzs < seg_len, torch.minimum(mn_ref, torch.zeros_like(mn_ref)), mn_ref)
    gm, _ = colstats(A, "min", inner=inner)
    if nseg:
        err = max(err, float((gm - mn_ref).abs().max()))
    if nrow * inner <= 400000:
        # rowSums over the leaves j = i + s * inner of every output column i (dims = 2 when inner > 1)
        rs = rowsums(A, inner=inner)
        leaf = torch.repeat_interleave(torch.arange(ncol, device=dev), torch.as_tensor(counts, device=dev))
        cell = (leaf % inner) * nrow + torch.as_tensor(ri, device=dev).long()
        r_ref = torch.zeros(nrow * inner, dtype=torch.float64, device=dev).index_add_(0, cell, vt)
        err = max(err, float((rs - r_ref).abs().max() / (1.0 + r_ref.abs().max())))
    torch.cuda.synchronize()
    worst = max(worst, err)
    flag = "" if err <= 1e-9 else "   <-- MISMATCH"
    print(f"{case:3d} nrow {nrow:7d} leaves {ncol:6d} inner {inner:3d} nnz {nnz:9d}  err {err:.2e}{flag}", flush=True)
    if flag:
        sys.exit(1)
print("worst error", worst)
