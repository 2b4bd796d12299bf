#!/bin/bash
for g in 32 64 16; do
  echo "== G=$g"; SVT_HIP_TUNING=1 SVT_SPMM_G=$g timeout -k 10 300 python tools/debug/config3_calls.py 20 2>&1 | grep "svt %"
done
SVT_HIP_TUNING=1 timeout -k 10 600 python tools/debug/fuzz_pbc.py 150 6001 2>&1 | tail -1
