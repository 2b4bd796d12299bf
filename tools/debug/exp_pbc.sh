#!/bin/bash
# Timing experiments on the DMA kernel (run on the GPU box).  EXPS: space-separated PBC_EXP
# values for tools/gen_pbc_asm.py, "base" = none, knobs joined by "_":
#   nofma  no index switch / FMA          nolds    no address adds / LDS reads of Y
#   nosmem no record loads (constant valid records, 7 batches per tile)
#   nodma  no LDS-DMA of Y                nocheck  finiteness result ignored (use with nodma)
#   nodmawait  no wait for the own DMA pieces   notouch, nofinite, noprio(removed), oldbound(removed)
#   nobarrier, align
# Experiment builds compute wrong results; the last line restores the real kernel.
# NEVER run nosmem without constant records (garbage register indices) -- the generator
# takes care of that.
for exp in ${EXPS:-base}; do
  [ "$exp" = base ] && exp=""
  PBC_EXP=$exp python tools/gen_pbc_asm.py > /dev/null && make -s -C sparsearray_amd/csrc 2>&1 | grep -E " error"
  echo "=== EXP='$exp'"
  timeout -k 10 200 python tools/tune_pbc.py --cfgs "${CFG:-40,16,7}" ${TUNE_ARGS:---reps 5} 2>&1 | grep -E "cfg|w00|w15"
done
PBC_EXP= python tools/gen_pbc_asm.py > /dev/null && make -s -C sparsearray_amd/csrc 2>&1 | grep -E " error"
