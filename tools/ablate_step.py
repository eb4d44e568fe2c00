"""What a family of launches costs the timed step: the default bench with that family's launches skipped (results are then wrong -- this is a
timing experiment, never a product mode).  python tools/ablate_step.py wgrad|none [bench args]
  wgrad : every weight-gradient launch (conv kernels + slab reduce + bias column sums) is skipped -> the step the main chain would run alone
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
what = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench  # noqa: E402
import viai_amd.ops as ops  # noqa: E402

if what == "wgrad":
    ops._wgrad_call = lambda *a, **k: None
elif what != "none":
    raise SystemExit("unknown ablation " + what)
bench.main()
