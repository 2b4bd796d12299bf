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
                      device="cuda", chunk_cols: int | None = None):
    """Returns (col_ptr int64[ncol+1], row_idx int32[nnz], val f64[nnz]) on
    ``device``.  Built in column chunks to bound peak memory: the number of
    nonzeros of a chunk is its exact share of floor(nrow*ncol*density)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
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


def random_dense(nrow: int, K: int, seed: int, device="cuda") -> torch.Tensor:
    """U(-1, 1) doubles; shape (K, nrow) C-contiguous == column-major nrow x K."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.rand((K, nrow), generator=g, device=device, dtype=torch.float64) * 2 - 1
