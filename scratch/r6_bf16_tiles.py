"""round 6: the bf16 operating point under forced grouped-GEMM tiles (butd_gemm_set_tile): does the bf16 instantiation -- whose
matrix time is ~nothing -- want other tiles than the fp32 rules choose?  python r6_bf16_tiles.py TM TN [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tm, tn = int(sys.argv[1]), int(sys.argv[2])
sys.argv = ["bench.py", "--dtype", "bf16", "--steps", "40", "--warmup", "5", "--no-extras", "--no-cpu-baseline"] + sys.argv[3:]
os.environ["BUTD_BENCH_NO_CHILD"] = "1"
from butd_detr_amd import _hiplib
assert _hiplib.load().butd_gemm_set_tile(tm, tn) == 0
import bench
bench.main()
