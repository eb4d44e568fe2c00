import sys, runpy, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from viai_amd import networks
networks.FLOW_STREAM = False
sys.argv = ["bench.py", "--config", "av", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-extra"]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
