#!/bin/bash
# one or two sets of y registers in the DMA kernel's record loop (run on the GPU box)
for ys in ${YS:-1 2 1 2}; do  # (the last line restores the default)
  PBC_YSETS=$ys python tools/gen_pbc_asm.py > /dev/null && make -s -C sparsearray_amd/csrc 2>&1 | grep -E " error"
  echo "=== YSETS=$ys"
  timeout -k 10 200 python tools/tune_pbc.py --cfgs "40,16,7" --reps 20 2>&1 | grep -E "cfg"
  [ "$ys" = 2 ] && timeout -k 10 200 python -m pytest tests/test_hip_device_level.py -x -q -m gpu 2>&1 | tail -1
done
python tools/gen_pbc_asm.py > /dev/null && make -s -C sparsearray_amd/csrc 2>&1 | grep -E " error"
