"""The shape of bench.py's contract line, without a GPU: the stdout line is a bounded projection of the detail record (round 5's 38 KB line
was not parsed by the driver)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for k, v in o.items():
            yield k
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


def test_contract_line_projection_is_small_for_any_record():
    """compact_line on a worst-case record (every optional block present, long strings everywhere) stays under the limit"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("viai_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    long = "x" * 5000
    fams = {"family_number_%02d_f16x2" % i: 0.123 for i in range(40)}
    leg = {"value": 1.0, "unit": "clips/s", "ms_per_step": 1.0, "steps": 3, "config": {"workload": long}, "roofline": {"bound": "mfma", "frac": 0.5, "kernel": long, "note": long},
           "cpu_baseline": {"value": 1.0, "unit": "clips/s", "cores": 16, "kind": "port", "sample": long}}
    full = {"metric": b.METRIC, "value": 1.0, "unit": "clips/s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": long, "math": long, "launch": long, "note": long},
            "roofline": {"bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None, "kernel": long, "family": "f", "frac_by_kernel": fams,
                         "conv_time_share_by_kernel": fams, "kernels": {k: long for k in fams}, "peak_note": long, "step_floor": {"model": long, "largest_gaps": [long] * 10}},
            "stages": {"note": long, "large_batch": {"note": long}}, "extra": {"av": leg, "wavenet": leg, "audio_exact_fp32": leg, "failed": {"error": long}},
            "cpu_baseline": {"value": 1.0, "unit": "clips/s", "cores": 16, "kind": "port", "sample": long}}
    line = b.compact_line(full)
    assert len(line) < 8000 and max(len(s) for s in _strings(json.loads(line))) <= 120
