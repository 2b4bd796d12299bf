#!/bin/bash
# gather kernel at BASELINE config 4: row-chunk length and splits per launch (tuning build)
export SVT_HIP_TUNING=1
for w in ${WAVES:-2048 4096 8192}; do for c in ${CHUNKS:-64 128 256 512}; do
  echo "== SVT_PBG_WAVES=$w SVT_PBG_CHUNK=$c"
  SVT_PBG_WAVES=$w SVT_PBG_CHUNK=$c timeout -k 10 300 python tools/debug/config4_time.py gather 2>&1 | grep "crossprod gather"
done; done
