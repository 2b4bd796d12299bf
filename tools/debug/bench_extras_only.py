"""bench.py with its extras and without the CPU baseline (for tools/debug/prof_py.sh: every kernel of the extras in one trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
sys.path.insert(0, ROOT)
exec(compile(open(os.path.join(ROOT, "bench.py")).read(), os.path.join(ROOT, "bench.py"), "exec"))
