"""Synthetic inputs with the distribution of the reference's
``randomSparseArray()`` (R/randomSparseArray.R:11-38): exactly
``floor(prod(dim) * density)`` nonzeros at positions drawn uniformly without
replacement, values ``signif(rnorm(n), 2)``.  R's RNG streams are not
reproduced (no R here); torch's Philox generator with a fixed seed is used, on
the device, so benchmark inputs are born in HBM.
"""
from __future__ import annotations

import torch


def _signif2(x: torch.Tensor) -> torch.Tensor:
    mag = torch.floor(torch.log10(torch.abs(x).clamp_min(1e-300)))
    f = torch.pow(10.0, 1.0 - mag)
    out = torch.round(x * f) / f
    out[out == 0] = 0.01
    return out


def random_device_csc(nrow: int, ncol: int, density: float, seed: int,
                      device="cuda", chunk_cols: int | None = None, total_nnz: int | None = None):
    """Returns (col_ptr int64[ncol+1], row_idx int32[nnz], val f64[nnz]) on
    ``device``.  Built in column chunks to bound peak memory: the number of
    nonzeros of a chunk is its exact share of floor(nrow*ncol*density)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if total_nnz is None:
        total_nnz = int(nrow * ncol * density)
    if chunk_cols is None:
        # ~5e7 candidate positions per chunk
        chunk_cols = max(1, min(ncol, int(5e7 / max(nrow * density, 1))))
    rows, vals, counts = [], [], []
    done_cols, done_nnz = 0, 0
    while done_cols < ncol:
        nc = min(chunk_cols, ncol - done_cols)
        want = (total_nnz * (done_cols + nc)) // ncol - done_nnz
        cells = nrow * nc
        if want > 0:
            m = int(want * 1.03) + 1024
            lin = torch.randint(0, cells, (m,), generator=g, device=device, dtype=torch.int64)
            lin = torch.unique(lin)            # sorted
            while lin.numel() < want:
                extra = torch.randint(0, cells, (m,), generator=g, device=device, dtype=torch.int64)
                lin = torch.unique(torch.cat([lin, extra]))
            if lin.numel() > want:             # drop the surplus uniformly
                drop = torch.randperm(lin.numel(), generator=g, device=device)[: lin.numel() - want]
                keep = torch.ones(lin.numel(), dtype=torch.bool, device=device)
                keep[drop] = False
                lin = lin[keep]
            col = torch.div(lin, nrow, rounding_mode="floor")
            rows.append((lin - col * nrow).to(torch.int32))
            counts.append(torch.bincount(col, minlength=nc))
            v = torch.randn(lin.numel(), generator=g, device=device, dtype=torch.float64)
            vals.append(_signif2(v))
            del lin, col, v
        else:
            counts.append(torch.zeros(nc, dtype=torch.int64, device=device))
        done_cols += nc
        done_nnz += max(want, 0)
    col_ptr = torch.zeros(ncol + 1, dtype=torch.int64, device=device)
    col_ptr[1:] = torch.cumsum(torch.cat(counts), 0)
    row_idx = torch.cat(rows) if rows else torch.zeros(0, dtype=torch.int32, device=device)
    val = torch.cat(vals) if vals else torch.zeros(0, dtype=torch.float64, device=device)
    return col_ptr, row_idx, val


def row_blocks(nrow: int, nblocks: int, align: int = 128):
    """The row cuts of the blocked generator: nblocks contiguous blocks, boundaries on multiples
    of `align` rows (the product kernel's row panels)."""
    units = (nrow + align - 1) // align
    base, rem = divmod(units, nblocks)
    cuts = [0]
    for b in range(nblocks):
        cuts.append(min(nrow, (b * base + min(b, rem) + base + (1 if b < rem else 0)) * align))
    return cuts


def random_device_csc_blocked(nrow: int, ncol: int, density: float, seed: int, device="cuda",
                              nblocks: int = 8, first: int = 0, last: int | None = None):
    """The same global matrix whatever the number of ranks: it is DEFINED as `nblocks` row
    blocks (cuts of row_blocks()), block b drawn on its own with seed (seed, b) and holding its
    exact share of floor(nrow*ncol*density) nonzeros.  Returns blocks first .. last-1 merged into
    one CSC (rows rebased to the first block's first row) and that row range.  A rank of an
    N-GPU run takes nblocks/N consecutive blocks; one GPU takes all of them.  (Stratified over
    the blocks, uniform inside each: the per-leaf counts are Binomial as in randomSparseArray().)"""
    if last is None:
        last = nblocks
    cuts = row_blocks(nrow, nblocks)
    total = int(nrow * ncol * density)
    parts = []
    for b in range(first, last):
        nb_rows = cuts[b + 1] - cuts[b]
        want = (total * (b + 1)) // nblocks - (total * b) // nblocks
        cp, ri, v = random_device_csc(nb_rows, ncol, density, seed * 1000 + b, device=device, total_nnz=want)
        parts.append((cp, ri, v, cuts[b] - cuts[first]))
    if len(parts) == 1:
        cp, ri, v, _ = parts[0]
        return cp, ri, v, (cuts[first], cuts[last])
    counts = torch.stack([p[0][1:] - p[0][:-1] for p in parts])          # [blocks, ncol]
    col_ptr = torch.zeros(ncol + 1, dtype=torch.int64, device=device)
    col_ptr[1:] = torch.cumsum(counts.sum(0), 0)
    before = torch.cumsum(counts, 0) - counts                           # nonzeros of earlier blocks per leaf
    nnz = int(col_ptr[-1])
    row_idx = torch.empty(nnz, dtype=torch.int32, device=device)
    val = torch.empty(nnz, dtype=torch.float64, device=device)
    cols = torch.arange(ncol, device=device)
    for i, (cp, ri, v, roff) in enumerate(parts):
        col = torch.repeat_interleave(cols, counts[i])
        dest = (col_ptr[:-1] + before[i] - cp[:-1])[col] + torch.arange(ri.numel(), device=device)
        row_idx[dest] = ri + roff
        val[dest] = v
        del col, dest
    return col_ptr, row_idx, val, (cuts[first], cuts[last])


def random_dense(nrow: int, K: int, seed: int, device="cuda") -> torch.Tensor:
    """U(-1, 1) doubles; shape (K, nrow) C-contiguous == column-major nrow x K."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.rand((K, nrow), generator=g, device=device, dtype=torch.float64) * 2 - 1


def random_dense_blocked(nrow: int, K: int, seed: int, device="cuda", nblocks: int = 8,
                         first: int = 0, last: int | None = None) -> torch.Tensor:
    """Rows of blocks first .. last-1 of the global U(-1, 1) matrix that is defined block by
    block like random_device_csc_blocked(); shape (K, rows) C-contiguous."""
    if last is None:
        last = nblocks
    cuts = row_blocks(nrow, nblocks)
    parts = [random_dense(cuts[b + 1] - cuts[b], K, seed * 1000 + b, device) for b in range(first, last)]
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1).contiguous()
