"""Differential fuzzing of the panel-blocked crossprod (LDS-DMA kernel, default layout) against the
general gather kernel (whose sums follow the reference's order and which implements the reference's
NaN / Inf / NA rules in full) on the GPU: random shapes, densities, skewed columns, dense operand
widths, odd row counts, the dense operand given by columns or by rows, and a few non-finite entries
in it (the per-column fix-up).  Run on the GPU box:
    python tools/debug/fuzz_pbc.py [ncases] [seed]
With SVT_HIP_TUNING=1 (a tuning build of the library) the row-split count is forced as well."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import _hip
from sparsearray_amd.device import CrossprodPlan, DeviceCSC, PbcPlan, set_gather_pacing, set_spare_cus
lib = _hip.init()

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
worst = 0.0
for case in range(ncases):
    nrow = int(rng.choice([256, 257, 300, 383, 384, 385, 1000, 1279, 1280, 4097, 20000, 65537]))
    ncol = int(rng.choice([1, 2, 39, 40, 41, 79, 640, 641, 1000, 1283, 2600]))
    K = int(rng.choice([1, 2, 7, 63, 64, 65, 100, 128, 129, 200]))
    kind = rng.integers(0, 5)
    dens = float(rng.choice([0.0005, 0.003, 0.01, 0.05, 0.3]))
    # every fifth case: a shape the XCD-paced gather kernel takes (K a multiple of 128, >= 64 panels of >= 512 rows),
    # with its pacing knobs and the number of idle CUs drawn as well
    paced = case % 5 == 4
    if paced:
        nrow = int(rng.choice([65537, 70001, 140003]))
        ncol = int(rng.choice([41, 640, 1283]))
        K = int(rng.choice([128, 128, 256]))
        dens = float(rng.choice([0.0005, 0.003, 0.01]))
        set_gather_pacing(int(rng.choice([0, 1, 2, 1000000, -1])), int(rng.choice([1, 256])))
        set_spare_cus(int(rng.choice([0, 0, 96])))
    cols = []
    for j in range(ncol):
        d = dens
        if kind == 1 and j % 7 == 0: d = 0.0                  # empty columns
        if kind == 2 and j % 11 == 3: d = min(1.0, dens * 40)  # a few heavy columns (tiles of many batches)
        if kind == 3: d = dens * (j + 1) / ncol                # ramp
        if kind == 4 and j == ncol // 2: d = 1.0               # one fully dense column
        n = rng.binomial(nrow, d)
        cols.append(np.sort(rng.choice(nrow, size=n, replace=False)).astype(np.int32) if n else np.zeros(0, np.int32))
    cp = np.zeros(ncol + 1, dtype=np.int64)
    cp[1:] = np.cumsum([len(c) for c in cols])
    ri = np.concatenate(cols) if cp[-1] else np.zeros(0, np.int32)
    v = np.round(rng.normal(size=len(ri)), 3)
    v[v == 0] = 0.5
    A = DeviceCSC.from_host(nrow, cp, ri, v)
    y = rng.uniform(-1, 1, (K, nrow))
    ns = int(rng.choice([0, 0, 2, 3, 5, 8, 16]))        # 0: automatic; else forced row-split count
    if hasattr(lib, "svt_dev_pbc_set_debug"):
        lib.svt_dev_pbc_set_debug(100 + ns)
    else:
        ns = 0
    npoison = int(rng.choice([0, 0, 0, 1, 2, 5]))
    for _ in range(npoison):
        r = int(rng.integers(0, nrow)) if rng.random() < 0.5 or len(ri) == 0 else int(ri[rng.integers(0, len(ri))])
        y[rng.integers(0, K), r] = rng.choice([np.inf, -np.inf, np.nan])
    # round 5: the classes of dirty dense columns the fix-up decides inside its leaf kernel (rounds 2-4: general kernels)
    pmode = int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
    if pmode == 1:                                           # many light columns: 1-3 entries in every second / every column
        for k in range(0, K, int(rng.choice([1, 2]))):
            for _ in range(int(rng.integers(1, 4))):
                r = int(rng.integers(0, nrow)) if rng.random() < 0.5 or len(ri) == 0 else int(ri[rng.integers(0, len(ri))])
                y[k, r] = rng.choice([np.inf, -np.inf, np.nan])
        npoison += K
    elif pmode == 2:                                         # one column between the list (256 entries) and a long leaf, or past it
        cnt = int(min(nrow, rng.choice([257, 300, 1000, 3000])))
        k = int(rng.integers(0, K))
        if rng.random() < 0.5 and len(cols) and len(max(cols, key=len)) >= cnt:
            y[k, max(cols, key=len)[:cnt]] = np.inf          # every entry on a nonzero of the longest leaf
        else:
            y[k, rng.choice(nrow, size=cnt, replace=False)] = rng.choice([np.inf, np.nan])
        npoison += cnt
    elif pmode == 3 and K >= 40:                             # more listed entries than the list holds (8192)
        per = min(nrow, 8192 // K + 40, 250)
        for k in range(K):
            y[k, rng.choice(nrow, size=per, replace=False)] = np.inf
        npoison += per * K
    elif pmode == 4:                                         # a whole column
        y[int(rng.integers(0, K)), :] = rng.choice([np.inf, np.nan])
        npoison += nrow
    Yd = torch.as_tensor(y, device=dev)
    by_rows = bool(rng.integers(0, 2))
    out_p = torch.full((K, ncol), 7.0, dtype=torch.float64, device=dev)
    out_g = torch.full((K, ncol), 9.0, dtype=torch.float64, device=dev)
    # layout: the default LDS-DMA one, the gather one (needs >= 512-row panels to make sense), or by density
    lay = [(40, 16, 7), (40, 4, 10), (0, 0, 0), (40, 4, 9)][int(rng.integers(0, 4))]
    if paced:
        lay = [(40, 4, 9), (40, 4, 10), (32, 4, 9), (16, 4, 10), (40, 4, 11 if nrow > 132000 else 10)][int(rng.integers(0, 5))]
    if by_rows:
        PbcPlan(A, K, *lay).run(Yd.t().contiguous(), K, out_p, tr_y=True)
    else:
        PbcPlan(A, K, *lay).run(Yd, nrow, out_p)
    CrossprodPlan(A, K).run(Yd, nrow, out_g)
    torch.cuda.synchronize()
    if paced:
        set_gather_pacing(); set_spare_cus(0)
    same_class = bool((torch.isnan(out_p) == torch.isnan(out_g)).all()) and \
        bool(((out_p == float("inf")) == (out_g == float("inf"))).all()) and \
        bool(((out_p == float("-inf")) == (out_g == float("-inf"))).all())
    fin = torch.isfinite(out_g) & torch.isfinite(out_p)
    # scale: sum of |a * y| per cell would be the honest one; the column's max |.| sum is a cheap bound
    scale = max(1.0, float(out_g[fin].abs().max())) if bool(fin.any()) else 1.0
    err = float((out_p - out_g)[fin].abs().max()) / scale if bool(fin.any()) else 0.0
    if not same_class:
        err = float("inf")
    worst = max(worst, err)
    flag = "" if err <= 1e-11 else "   <-- MISMATCH"
    print(f"{case:4d} nrow {nrow:6d} ncol {ncol:5d} K {K:4d} kind {kind} dens {dens:<7g} nsplit {ns:2d} poison {npoison:6d} mode {pmode} by_rows {int(by_rows)} layout {lay} nnz {len(ri):8d}  err {err:.2e}{flag}", flush=True)
    if flag:
        bad = ~(((out_p - out_g).abs() / scale <= 1e-11) | (torch.isnan(out_p) & torch.isnan(out_g)) | (out_p == out_g))
        idx = bad.nonzero()[:5].tolist()
        print("   first bad (k, col):", idx)
        sys.exit(1)
if hasattr(lib, "svt_dev_pbc_set_debug"):
    lib.svt_dev_pbc_set_debug(100)
print("worst relative error", worst)
