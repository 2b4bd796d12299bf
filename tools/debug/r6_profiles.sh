#!/bin/bash
# Round 6: the summaries that go to profiles/ (run on the GPU box from the repository root, as the LAST GPU job of the
# round: the files must describe the committed HEAD).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r6
mkdir -p $O
cd $ROOT
# 1. the bench command: kernel trace + FETCH_SIZE / WRITE_SIZE passes (tools/profile.sh), summary + traffic.json
bash tools/profile.sh r06 > $O/profile_r06.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_r06 $O/r06_bench > /dev/null 2>&1
echo "== bench summary"; head -30 $O/r06_bench_summary.txt
# 2. the bench line itself (with extras: the reference's published crossprod shapes, config-2-scale crossprod(A))
timeout -k 10 900 python3 bench.py > $O/r06_bench_line.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 600 $O/r06_bench_line.json
# 3. unary / SVT x SVT crossprod: events, then per-kernel averages and counters of the sparse-aware kernel
{
echo "==== tools/debug/sparse_crossprod_time.py all 5 (HIP events; the dense-buffer route in wall time)"
timeout -k 10 600 python3 tools/debug/sparse_crossprod_time.py all 5 2>&1 | grep -v amdgpu.ids
echo "==== rocprofv3 --kernel-trace --stats of tools/debug/gram_only.py 5 1 (crossprod(A), A 1e6 x 1e4 @ 1 %, symmetric form)"
bash tools/debug/prof_py.sh tools/debug/gram_only.py 3 "5 1" 2>&1 | grep -v amdgpu.ids | grep -i "gram\|transpose\|scan_"
echo "==== rocprofv3 --pmc, one pass per line of tools/debug/pmc_sets_gram.txt, kernel gram_sym_kernel (per-launch averages; FETCH_SIZE / WRITE_SIZE in KiB as reported)"
bash tools/debug/pmc_py.sh tools/debug/gram_only.py gram_sym "2 1" tools/debug/pmc_sets_gram.txt 2>&1 | grep -v amdgpu.ids
echo "==== the same for the general form (crossprod(x, x) without the symmetry), kernel gram_gen_kernel"
printf 'FETCH_SIZE\nTCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum\n' > $O/sets2.txt
bash tools/debug/pmc_py.sh tools/debug/gram_only.py "gram_gen" "2 0" gpurun_out/r6/sets2.txt 2>&1 | grep -v amdgpu.ids
} > $O/r06_sparse_crossprod.txt 2>&1
tail -40 $O/r06_sparse_crossprod.txt
# 4. other kernels touched this round
{
echo "==== tools/debug/aperm4d_time.py (the key-sort route's pointer fill: c(3,2,4,1))"
timeout -k 10 300 python3 tools/debug/aperm4d_time.py 2>&1 | grep -v amdgpu.ids
bash tools/debug/prof_py.sh tools/debug/aperm4d_time.py 3 2>&1 | grep -v amdgpu.ids | grep -i "aperm\|sort\|radix\|transpose\|scan"
echo "==== tools/debug/nonfinite_and_rowmajor_time.py (dirty dense columns, incl. every column in the walked class)"
timeout -k 10 300 python3 tools/debug/nonfinite_and_rowmajor_time.py 2>&1 | grep -v amdgpu.ids
echo "==== tools/debug/config3_calls.py"
timeout -k 10 300 python3 tools/debug/config3_calls.py 2>&1 | grep -v amdgpu.ids
echo "==== tools/debug/config2b_time.py"
timeout -k 10 300 python3 tools/debug/config2b_time.py 2>&1 | grep -v amdgpu.ids
echo "==== tools/debug/fuzz_transpose.py 150 61 (which routes random shapes take)"
timeout -k 10 500 python3 tools/debug/fuzz_transpose.py 150 61 2>&1 | grep -v amdgpu.ids | tail -2
} > $O/r06_other_kernels.txt 2>&1
tail -30 $O/r06_other_kernels.txt
# 5. RCCL with one rank (what a one-GPU box can show: the library loads, the collective path runs, results are those without a group)
timeout -k 10 400 python3 tests/workers/rccl_one_rank_worker.py $O/r06_rccl_one_rank.json > $O/rccl.log 2>&1
echo "rccl rc=$?"; cat $O/r06_rccl_one_rank.json
