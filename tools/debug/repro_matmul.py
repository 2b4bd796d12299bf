"""Reproducer for svt_matmul_SVT_SVT at the size of tests/test_hip_vs_oracle.py::test_matmul_one_call_large."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import random_csc
import sparsearray_amd as sa
import sparsearray_amd._hip as _h
if os.environ.get("SVT_LIB"):
    _h.LIB_PATH = os.environ["SVT_LIB"]
    print("using", _h.LIB_PATH, flush=True)
from sparsearray_amd import SVT_SparseArray
import scipy.sparse as sp
hip = sa.hip_session() if hasattr(sa, "hip_session") else sa.session()
cp, ri, v = random_csc(2000, 300_000, 0.01, 21)
xt = sp.csc_matrix((v, ri, cp), shape=(2000, 300_000))
t = xt.T.tocsc(); t.sort_indices()
x = SVT_SparseArray.from_csc((300_000, 2000), "double", t.indptr.astype(np.int64), t.indices.astype(np.int32), t.data)
print("x built", flush=True)
y = np.random.default_rng(22).uniform(-1, 1, (2000, 70))
r1 = hip.matmul(x, y)
print("dense ok", r1.shape, flush=True)
cpb, rib, vb = random_csc(2000, 50, 0.05, 23)
b = SVT_SparseArray.from_csc((2000, 50), "double", cpb, rib, vb)
r2 = hip.matmul(x, b)
print("sparse ok", r2.shape, flush=True)
