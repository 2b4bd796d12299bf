#!/bin/bash
# timing experiments on the DMA kernel's record loop (run on the GPU box)
for exp in ${EXPS:-"" double}; do
  [ "$exp" = base ] && exp=""
  PBC_EXP=$exp python tools/gen_pbc_asm.py > /dev/null && make -s -C sparsearray_amd/csrc 2>&1 | grep -E "error" 
  echo "=== EXP='$exp'"
  timeout -k 10 200 python tools/tune_pbc.py --cfgs "${CFG:-32,16,7}" --prof 2>&1 | grep -E "cfg|w00|w15"
done
PBC_EXP= python tools/gen_pbc_asm.py > /dev/null && make -s -C sparsearray_amd/csrc
