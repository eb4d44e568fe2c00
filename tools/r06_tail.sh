d=/tmp/tq_$$; mkdir -p $d; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $GRAFT_REPO_ROOT/bench.py --plan --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-extra > $d/log.txt 2>&1
cd $GRAFT_REPO_ROOT; python tools/step_queues.py $d 10 5 --tail 45 | tail -64
