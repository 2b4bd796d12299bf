#!/bin/bash
# round-2 profiles of the bench command: kernel trace + HBM counters + SQ counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/profile.sh r02 > $R/gpurun_out/prof_r02.log 2>&1
echo "profile.sh done" 
cd $R && python3 tools/summarize_profile.py gpurun_out/prof_r02 gpurun_out/r02_bench > /dev/null 2>&1
bash $R/tools/debug/pmc_pbc.sh > $R/gpurun_out/r02_sq_counters_raw.txt 2>&1
echo "pmc done"
tail -30 $R/gpurun_out/r02_sq_counters_raw.txt
cat $R/gpurun_out/r02_bench_summary.txt
