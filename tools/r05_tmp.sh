d=/tmp/avp; rm -rf $d; mkdir -p $d gpurun_out/r05_av
(cd /tmp && TMPDIR=/tmp VIAI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $GRAFT_REPO_ROOT/tools/r05_tmp2.py > $d/log.txt 2>&1)
f=$(find $d -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r05_av/av_serial4_kernel_stats.csv
grep '^{"metric"' $d/log.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("serial ms", d["ms_per_step"])'
