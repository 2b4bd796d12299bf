#!/bin/bash
# A/B of product-kernel variants built into side libraries (libsvt_hip_<tag>.so), same box, same process order
cd $GRAFT_REPO_ROOT
cp sparsearray_amd/libsvt_hip.so /tmp/base.so
for tag in ${TAGS:-rot0 rot1 rot5 rot0 rot1 rot5}; do
  cp sparsearray_amd/libsvt_hip_$tag.so sparsearray_amd/libsvt_hip.so
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag: ms/step %.4f kernel %.4f checksum %s' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['result_checksum']['abs_sum']))"
done
cp /tmp/base.so sparsearray_amd/libsvt_hip.so
