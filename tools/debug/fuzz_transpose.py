"""Differential fuzzing of the device transposition (bucketed count / scan / scatter, and its key-sort
fallback for the shapes it does not take) against a stable sort by row on the GPU: random shapes from one
row to 3e5, from one column to 2e5, densities from 1e-4 to 1, skewed rows and columns, both value types.
    python tools/debug/fuzz_transpose.py [ncases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd.device import DeviceCSC

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
bad = 0
for case in range(ncases):
    nrow = int(rng.choice([1, 2, 63, 64, 65, 255, 1000, 1024, 1025, 4097, 20000, 65537, 300000]))
    ncol = int(rng.choice([1, 2, 255, 256, 257, 1000, 5000, 40000, 200000]))
    dens = float(rng.choice([0.0001, 0.001, 0.01, 0.05, 0.3, 1.0]))
    if nrow * ncol * dens > 3e7:
        dens = 3e7 / (nrow * ncol)
    kind = int(rng.integers(0, 4))
    g = torch.Generator(device=dev); g.manual_seed(int(rng.integers(0, 2 ** 31)))
    # a dense mask would not fit: draw linear positions
    n_try = int(nrow * ncol * dens)
    if n_try <= 0:
        n_try = 1
    lin = torch.randint(0, nrow * ncol, (n_try,), generator=g, device=dev, dtype=torch.int64)
    if kind == 1:                                   # a band of heavy rows
        r0 = int(rng.integers(0, nrow))
        extra = torch.arange(ncol, device=dev, dtype=torch.int64) * nrow + r0
        lin = torch.cat([lin, extra])
    if kind == 2:                                   # a few fully dense columns
        for c0 in rng.integers(0, ncol, size=min(3, ncol)):
            lin = torch.cat([lin, int(c0) * nrow + torch.arange(nrow, device=dev, dtype=torch.int64)])
    if kind == 3:                                   # rows crowded at the low end
        lin = (lin // nrow) * nrow + (lin % nrow) % max(1, nrow // 7)
    lin = torch.unique(lin)                         # sorted: column-major order
    ri = (lin % nrow).to(torch.int32)
    col = lin // nrow
    cp = torch.zeros(ncol + 1, dtype=torch.int64, device=dev)
    cp[1:] = torch.cumsum(torch.bincount(col, minlength=ncol), 0)
    is_int = bool(rng.integers(0, 2))
    if is_int:
        v = torch.randint(-50, 50, (lin.numel(),), generator=g, device=dev, dtype=torch.int32)
    else:
        v = torch.randn(lin.numel(), generator=g, device=dev, dtype=torch.float64)
    A = DeviceCSC(nrow, cp, ri, v)
    # every third case with columns that split into slabs: the operand read as an nrow x d2 x nslab array and
    # aperm(x, c(2, 1, 3)) -- the batched form of the bucketed transposition -- and the two permutations composed from it,
    # against a sort of the permuted indices
    nslab = next((q for q in (7, 5, 4, 3, 2) if ncol % q == 0 and ncol // q >= 1), 0) if case % 3 == 2 else 0
    if nslab:
        d2 = ncol // nslab
        j, sl = col % d2, col // d2
        subs, dims3 = [ri.to(torch.int64), j, sl], (nrow, d2, nslab)
        for perm in ((2, 1, 3), (2, 3, 1), (3, 2, 1)):
            if dims3[perm[1] - 1] * dims3[perm[2] - 1] > 5e7:          # (new leaves: their pointers alone would be GBs)
                continue
            P, new_dim = A.aperm(dims3, perm)
            torch.cuda.synchronize()
            ns = [subs[q - 1] for q in perm]
            nd = tuple(dims3[q - 1] for q in perm)
            key = ns[0] + nd[0] * (ns[1] + nd[1] * ns[2])                # linear index in the permuted array
            order = torch.sort(key, stable=True).indices
            want_cp = torch.zeros(nd[1] * nd[2] + 1, dtype=torch.int64, device=dev)
            want_cp[1:] = torch.cumsum(torch.bincount(ns[1] + nd[1] * ns[2], minlength=nd[1] * nd[2]), 0)
            ok3 = new_dim == nd and torch.equal(P.col_ptr, want_cp) and \
                torch.equal(P.row_idx, ns[0][order].to(torch.int32)) and torch.equal(P.val, v[order])
            if not ok3:
                bad += 1
                print(f"MISMATCH (aperm {perm}) case {case}: dim {dims3} nnz {lin.numel()} kind {kind} int {is_int}", flush=True)
            del P, key, order, want_cp, ns
        del j, sl, subs
    # round 5: every fourth case the columns read as three or four further axes and a RANDOM permutation of all axes (the
    # composed route: leaf-preserving step, first two axes swapped or slab form, leaf-preserving step; or the own radix sort)
    if case % 4 == 1 and ncol >= 8:
        facs, rest = [], ncol
        for q in (2, 3, 5, 7, 2, 3):
            if rest % q == 0 and rest // q >= 1 and len(facs) < 3:
                facs.append(q); rest //= q
        if len(facs) >= 2:
            dimsN = (nrow, rest) + tuple(facs)
            strides, st = [], 1
            for dsz in dimsN[1:]:
                strides.append(st); st *= dsz
            subsN = [ri.to(torch.int64)] + [(col // stv) % dsz for stv, dsz in zip(strides, dimsN[1:])]
            for _ in range(3):
                perm = tuple(int(q) + 1 for q in rng.permutation(len(dimsN)))
                nd = tuple(dimsN[q - 1] for q in perm)
                nl = int(np.prod(nd[1:], dtype=np.int64))
                if nl > 5e7:
                    continue
                P, new_dim = A.aperm(dimsN, perm)
                torch.cuda.synchronize()
                ns = [subsN[q - 1] for q in perm]
                leafN = torch.zeros_like(ns[0]); mul = 1
                for a in range(1, len(nd)):
                    leafN = leafN + ns[a] * mul; mul *= nd[a]
                order = torch.sort(ns[0] + nd[0] * leafN, stable=True).indices
                want_cp = torch.zeros(nl + 1, dtype=torch.int64, device=dev)
                want_cp[1:] = torch.cumsum(torch.bincount(leafN, minlength=nl), 0)
                okN = new_dim == nd and torch.equal(P.col_ptr, want_cp) and \
                    torch.equal(P.row_idx, ns[0][order].to(torch.int32)) and torch.equal(P.val, v[order])
                if not okN:
                    bad += 1
                    print(f"MISMATCH (aperm {perm}) case {case}: dim {dimsN} nnz {lin.numel()} kind {kind} int {is_int}", flush=True)
                del P, order, want_cp, leafN, ns
            del subsN
    T = A.t()
    torch.cuda.synchronize()
    order = torch.sort(ri.to(torch.int64), stable=True).indices
    want_idx = col[order].to(torch.int32)
    want_val = v[order]
    want_cp = torch.zeros(nrow + 1, dtype=torch.int64, device=dev)
    want_cp[1:] = torch.cumsum(torch.bincount(ri.to(torch.int64), minlength=nrow), 0)
    ok = torch.equal(T.col_ptr, want_cp) and torch.equal(T.row_idx, want_idx) and torch.equal(T.val, want_val)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: nrow {nrow} ncol {ncol} nnz {lin.numel()} kind {kind} int {is_int}", flush=True)
    del A, T, lin, ri, col, cp, v, order, want_idx, want_val, want_cp
print(f"{ncases} cases, {bad} mismatches")
from sparsearray_amd.device import aperm_route_counts
rc = aperm_route_counts()
tot = max(1, sum(rc.values()))
print("routes taken (steps of composed routes included): " + ", ".join(f"{k} {v}" for k, v in rc.items())
      + f"; key sorts = {100.0 * (rc['t_key_sort'] + rc['key_sort_32'] + rc['key_sort_64']) / tot:.1f} % of {tot} route steps")
sys.exit(1 if bad else 0)
