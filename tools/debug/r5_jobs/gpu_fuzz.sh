#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 700 python tools/debug/fuzz_pbc.py 400 5001 > gpurun_out/r5/fuzz_pbc.log 2>&1; echo "pbc rc=$?"; tail -2 gpurun_out/r5/fuzz_pbc.log; grep -c MISMATCH gpurun_out/r5/fuzz_pbc.log
timeout -k 10 300 python tools/debug/fuzz_transpose.py 150 5002 > gpurun_out/r5/fuzz_t.log 2>&1; echo "t rc=$?"; tail -1 gpurun_out/r5/fuzz_t.log
timeout -k 10 300 python tools/debug/fuzz_spmm.py 150 5003 > gpurun_out/r5/fuzz_s.log 2>&1; echo "spmm rc=$?"; tail -1 gpurun_out/r5/fuzz_s.log
timeout -k 10 300 python tools/debug/fuzz_stats.py 60 5004 > gpurun_out/r5/fuzz_st.log 2>&1; echo "stats rc=$?"; tail -1 gpurun_out/r5/fuzz_st.log
