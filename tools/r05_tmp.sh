timeout 900 python -m pytest tests/test_resnet_gpu.py -x -q 2>&1 | tail -4
for i in 1 2; do python bench.py --config av --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("av ms", d["ms_per_step"], d["roofline"].get("traffic") if "roofline" in d else None)'; done
