"""crossprod(x) with a result taller than one workgroup's LDS (more than 10 200 columns): the cell-panel form of the
sparse-aware kernel against the dense-buffer route.   python tools/debug/wide_result_time.py"""
import os, sys, torch
ROOT0 = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT0)
sys.argv = ["x", "none"]
exec(open(os.path.join(ROOT0, "tools", "debug", "sparse_crossprod_time.py")).read().split('if what in ("small", "all"):')[0])
from sparsearray_amd.device import set_sparse_crossprod_panel
for n_, r_, d_, seed_ in ((20_000, 100_000, 0.01, 21), (12_000, 200_000, 0.005, 22)):
    unary(r_, n_, d_, seed_, with_dense=True, with_spmm=False, tag=" [defaults: one block up to 16 384 (symmetric) / 20 400 (general) columns, panels of 8192 beyond]")
    set_sparse_crossprod_panel(10200, 13)
    unary(r_, n_, d_, seed_, with_dense=False, with_spmm=False, tag=" [the same by panels of 8192 cells]")
    set_sparse_crossprod_panel(20400, 13)
    unary(r_, n_, d_, seed_, with_dense=False, with_spmm=False, tag=" [the same in one block whatever the form]") if n_ <= 16384 else None
    set_sparse_crossprod_panel(-1, -1)
for ps_ in (13, 14):
    set_sparse_crossprod_panel(-1, ps_)
    unary(100_000, 30_000, 0.005, 23, with_dense=(ps_ == 13), with_spmm=False, tag=f" [30 000 columns: panels of {1 << ps_} cells]")
set_sparse_crossprod_panel(-1, -1)
