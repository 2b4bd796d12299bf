#!/bin/bash
# round-3 profiles of the bench command: kernel trace + HBM counters (tools/profile.sh) + SQ counters of the product kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/profile.sh r03 > $R/gpurun_out/prof_r03.log 2>&1
echo "profile.sh done"
cd $R && python3 tools/summarize_profile.py gpurun_out/prof_r03 gpurun_out/r03_bench > /dev/null 2>&1
bash $R/tools/debug/pmc_pbc.sh > $R/gpurun_out/r03_sq_counters_raw.txt 2>&1
echo "pmc done"
tail -28 $R/gpurun_out/r03_sq_counters_raw.txt
cat $R/gpurun_out/r03_bench_summary.txt
