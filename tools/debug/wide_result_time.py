"""crossprod(x) with a result taller than one workgroup's LDS (more than 10 200 columns): the cell-panel form of the
sparse-aware kernel against the dense-buffer route.   python tools/debug/wide_result_time.py"""
import os, sys, torch
ROOT0 = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT0)
sys.argv = ["x", "none"]
exec(open(os.path.join(ROOT0, "tools", "debug", "sparse_crossprod_time.py")).read().split('if what in ("small", "all"):')[0])
unary(100_000, 20_000, 0.01, 21, with_dense=True, with_spmm=False, tag=" [wide result: cell panels]")
unary(200_000, 12_000, 0.005, 22, with_dense=True, with_spmm=False, tag=" [wide result: cell panels]")
