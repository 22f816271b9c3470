"""bench.py with the decode GEMV's grid forced: bench_tuned.py BLOCKS WAVES [bench args...]  (0 0 = automatic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # first: torch brings its own HIP runtime, the library must bind to the same one
import fastllm_amd as fa
fa.tune("gemv_blocks", int(sys.argv[1])); fa.tune("gemv_waves", int(sys.argv[2]))
sys.argv = ["bench.py"] + sys.argv[3:]
import bench
bench.main()
