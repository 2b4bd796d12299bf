#!/bin/bash
# Round 5: the summaries that go to profiles/ (run on the GPU box from the repository root).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r5
mkdir -p $O
cd $ROOT
bash tools/profile.sh r05 > $O/profile_r05.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_r05 $O/r05_bench > /dev/null 2>&1
echo "== bench summary"; head -24 $O/r05_bench_summary.txt
{
for sc in config2b_time.py config3_calls.py median_time.py aperm_time.py aperm4d_time.py; do
  echo "==== tools/debug/$sc (events, then rocprofv3 --kernel-trace --stats per-kernel averages)"
  timeout -k 10 300 python3 tools/debug/$sc 2>&1 | grep -v amdgpu.ids
  bash tools/debug/prof_py.sh tools/debug/$sc 3 2>&1 | grep -v "amdgpu.ids"
done
} > $O/r05_other_kernels.txt 2>&1
tail -50 $O/r05_other_kernels.txt
timeout -k 10 400 python3 bench.py --config 4 --nrow 1250000 --steps 10 --warmup 2 --no-extras > $O/r05_config4_rank_share.json 2> $O/c4a.err
echo "c4 share rc=$?"; tail -c 900 $O/r05_config4_rank_share.json
timeout -k 10 600 python3 bench.py --config 4 --steps 5 --warmup 2 --no-extras > $O/r05_config4_one_gpu.json 2> $O/c4b.err
echo "c4 one gpu rc=$?"; tail -c 900 $O/r05_config4_one_gpu.json
