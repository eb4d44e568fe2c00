R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_wavenet_gpu.py -x -q 2>&1 | tail -3
python $R/tools/wn_synth.py 2048 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wn -o wn --output-format csv -- python $R/tools/wn_synth.py 512 > /dev/null 2>&1
rm -f $R/gpurun_out/wn/wn_kernel_trace.csv
python - <<'P'
import csv,os
for r in list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/wn/wn_kernel_stats.csv')))[:5]:
    print(r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])
P
