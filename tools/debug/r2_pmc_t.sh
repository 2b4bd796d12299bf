#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_t_$tag -- python3 $R/tools/debug/t_only.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_t_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gather" in n or "onesweep" in n or "iota" in n or "bounds" in n:
        agg[(n[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k[0], k[1], "avg %.0f" % (sum(v) / len(v)), "n", len(v))
PY
done
