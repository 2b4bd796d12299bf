#!/bin/bash
for sp in 4 8 16; do
  echo "== subp $sp"; SVT_HIP_TUNING=1 SVT_PBC_SUBP=$sp timeout -k 10 300 python tools/debug/build_time.py 2>&1 | grep -v amdgpu
done
