VIAI_PLAN_DEBUG=1 timeout 600 python bench.py --config av --plan --steps 5 --warmup 2 2>&1 | grep -v "^\[W\|amdgpu.ids" | cut -c1-250 | tail -8
timeout 600 python bench.py --config av --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['launch_modes'])"
