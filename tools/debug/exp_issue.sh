#!/bin/bash
# DMA issue modes of the crossprod kernel (run on the GPU box)
for mode in "pieces 1" "pieces 2" "stagger 1"; do
  set -- $mode
  PBC_ISSUE=$1 PBC_NPP=$2 python tools/gen_pbc_asm.py > /dev/null && make -s -C sparsearray_amd/csrc 2>&1 | grep -E " error"
  echo "=== issue=$1 npp=$2"
  timeout -k 10 200 python tools/debug/dma_check.py 2>&1 | grep -E "rep=0" | awk '{print $1,$2,$4,$6,$7,$8}' | tr '\n' ';'; echo
  timeout -k 10 200 python tools/tune_pbc.py --cfgs "40,16,7;32,16,7" --prof 2>&1 | grep -E "cfg|w00|w15"
done
