# round-6 evidence run: bench line (the driver's command line), kernel stats + queue report of the three-stream step, single-stream timeline, counters of the step and of the
# dominant kernel's layer, the other configs' kernel stats, the WaveNet pipeline's counters and stage stamps.   bash tools/r06_final.sh <tag>
tag=$1; o=gpurun_out/r06_$tag; mkdir -p $o
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench.log 2> $o/bench.err; tail -1 $o/bench.log > $o/bench_line.json; cp gpurun_out/bench_detail_audio.json $o/bench_detail.json
tools/step_profile.sh $o r06_$tag > $o/step_profile.log 2>&1
tools/pmc_step.sh $o/pmc_step.json > $o/pmc_step.log 2>&1
tools/pmc_layer.sh $o/pmc_dconv3.json --shape 16 64 32 256 512 --bn --p16 > /dev/null 2>&1
d=/tmp/avp_$$; rm -rf $d; mkdir -p $d
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python "$GRAFT_REPO_ROOT/bench.py" --config av --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extra > $d/log.txt 2>&1)
f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $o/av_kernel_stats.csv
bash tools/prof_stats.sh r06_${tag}_wn --config wavenet --no-cpu-baseline > $o/wn_stats.log 2>&1; cp gpurun_out/r06_${tag}_wn_kernel_stats.csv $o/wn_kernel_stats.csv
bash tools/pmc_script.sh $o/pmc_wn_pipe.json bench.py --config wavenet --no-cpu-baseline --steps 1024 --warmup 256 > $o/pmc_wn.log 2>&1
python tools/wn_pipe_stamps.py 700 2>/dev/null | grep -v Warn > $o/wn_pipe_stamps.txt
tail -3 $o/pmc_step.log; python -c "
import json; d=json.load(open('$o/bench_line.json')); print(len(open('$o/bench_line.json').read()), d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('step_floor_ms'), d['roofline'].get('frac_of_floor'), {k: (v.get('ms_per_step'), v.get('value')) for k, v in d['extra'].items()})"
