#!/bin/bash
# A/B of product-kernel variants built into side libraries (libsvt_hip_<tag>.so), same box, same process order
#   TAGS="a b a b" [EXTRAS=1] bash tools/debug/r2_rot.sh      (EXTRAS=1: also the A %*% Y time of the bench extras)
cd ${GRAFT_REPO_ROOT:-.}
cp sparsearray_amd/libsvt_hip.so /tmp/base.so
for tag in ${TAGS:-rot0 rot1 rot5 rot0 rot1 rot5}; do
  cp sparsearray_amd/libsvt_hip_$tag.so sparsearray_amd/libsvt_hip.so
  if [ -n "$EXTRAS" ]; then X=""; else X="--no-extras"; fi
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $X 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); e=j.get('extras') or {}; print('$tag: ms/step %.4f kernel %.4f checksum %s  A%%*%%Y %s' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['result_checksum']['abs_sum'], e.get('matmul_A_Y(2b)', {}).get('ms')))"
done
cp /tmp/base.so sparsearray_amd/libsvt_hip.so
