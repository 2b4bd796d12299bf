#!/bin/bash
# The CPU-side code of the repository under AddressSanitizer + UndefinedBehaviorSanitizer (build container; GPU ASan is not
# available on the pool): the oracle and the executed R glue (integration/svt_hip_glue.c with the functional R stand-in,
# the reference helper files from the read-only mount and the svt_* -> oracle shim), driven with every golden case x
# {lacunar, plain}.  Round 5: 3804 case runs, 2098 .Call()s through the glue, no report.
#   bash tools/debug/asan_cpu.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; REF=/root/reference/src; O=/tmp/svt_asan; mkdir -p $O; cd $O
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g -fPIC"
gcc $SAN -std=gnu99 -fopenmp -ffp-contract=off -shared -o libsvt_oracle.so $R/oracle/svt_oracle.c -lm
for f in argcheck_utils Rvector_utils Rvector_summarization; do
  gcc -c $SAN -fvisibility=hidden -ffunction-sections -fdata-sections -w -I$R/tests/r_api_standin -I$REF $REF/$f.c -o $f.o
done
gcc -c $SAN -I$R/tests/r_api_standin -I$REF -I$R/include $R/integration/svt_hip_glue.c -o glue.o
gcc -c $SAN -I$R/tests/r_api_standin $R/tests/r_api_standin/r_standin.c -o r_standin.o
gcc -c $SAN -w -I$R/tests/r_api_standin -I$REF $R/tests/r_api_standin/glue_env.c -o glue_env.o
gcc -shared -fsanitize=address,undefined -o libglue_harness.so glue.o r_standin.o glue_env.o argcheck_utils.o Rvector_utils.o \
    Rvector_summarization.o -Wl,--gc-sections -L. -lsvt_oracle -Wl,-rpath,$O -ldl -lm
gcc -shared $SAN -I$R/include -I$R/oracle $R/tests/r_api_standin/svt_over_oracle.c -o libsvt_shim.so -L. -lsvt_oracle -Wl,-rpath,$O
cat > run.py <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import oracle.oracle as oo
oo._LIB = "$O/libsvt_oracle.so"
from oracle.oracle import oracle_dispatcher
from sparsearray_amd.api import Session
import glue_harness
from helpers import check_case, golden_cases
g = glue_harness.Glue("$O/libglue_harness.so", "$O/libsvt_shim.so")
sess_glue = Session(glue_harness.GlueDispatcher(g, oracle_dispatcher()))
sess_orc = Session(oracle_dispatcher())
n = 0
for case in golden_cases():
    for lac in (True, False):
        check_case(sess_orc, case, lacunar=lac)
        check_case(sess_glue, case, lacunar=lac)
        n += 2
print("cases run under ASan/UBSan:", n, "glue calls:", g.stats)
PY
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=2 python3 run.py
