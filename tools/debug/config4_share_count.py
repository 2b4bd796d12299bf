"""how many of the gather kernel's row fetches at BASELINE config 4 could a
workgroup-level LDS exchange save?  Counted on the operand's distribution (Bernoulli(0.001) cells, 2048-row panels,
40-column groups; CPU, no GPU needed): a row of Yt fetched by one wavefront serves the other wavefronts of the
WORKGROUP (the scope LDS is shared in) that hold a nonzero in the same row of the same panel.

Result (400 panels): 4 wavefronts per workgroup (the kernel's shape): 5.6 % of the fetches; all 8 wavefronts of a CU
(one 512-thread workgroup per CU instead of two of 256): 12.4 %; repeats inside one wavefront (L1 hits already): 1.9 %.
The criterion for building it was >= 10 % on the product (<= 3.85 ms at the rank's share) -- the ceiling of the
kernel's shape is 5.6 % of the bytes of a kernel that is 80 % bound by them, before any protocol cost: not built.
"""
import numpy as np

rng = np.random.default_rng(1)
dens, P, G = 0.001, 2048, 40
for W in (4, 8):
    tot = saved = own = 0
    for _ in range(400):
        m = rng.random((P, W * G)) < dens            # one panel x the columns of one workgroup (W wavefronts)
        per_group = m.reshape(P, W, G).sum(axis=2)   # records per (row, wavefront)
        need = (per_group > 0).sum(axis=1)           # wavefronts that fetch the row
        tot += int(m.sum())
        saved += int(np.maximum(need - 1, 0).sum())
        own += int(np.maximum(per_group - 1, 0).sum())
    print(f"{W} wavefronts share LDS: {tot} records, fetches another wavefront's fetch could serve: {saved} = {saved / tot:.4f}; "
          f"repeats inside a wavefront: {own / tot:.4f}")
