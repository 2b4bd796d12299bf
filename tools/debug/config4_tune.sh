#!/bin/bash
export SVT_HIP_TUNING=1
for w in 4096 8192; do for c in 128 256 512 1024; do
  echo "== SVT_PBG_WAVES=$w SVT_PBG_CHUNK=$c"
  SVT_PBG_WAVES=$w SVT_PBG_CHUNK=$c timeout -k 10 300 python tools/debug/config4_time.py gather 2>&1 | grep "crossprod gather"
done; done
