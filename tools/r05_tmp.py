import sys, runpy, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from viai_amd import ops, networks
mode = sys.argv[1]
if mode == "nolazy": ops.LAZY_SUM = False
if mode == "nojoin": ops.JOIN_FUSED = False
if mode == "notwin": networks.P16_TWIN = False
if mode == "pooladd": ops.POOL_ADDENDS = True
sys.argv = ["bench.py", "--config", "av", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-roofline", "--no-extra"]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
