# round-5 evidence run: bench line, kernel stats, timelines, counters.   bash tools/r05_final.sh <tag>
tag=$1; o=gpurun_out/r05_$tag; mkdir -p $o
python bench.py > $o/bench.log 2>&1; grep '^{"metric"' $o/bench.log > $o/bench_line.json
tools/step_profile.sh $o r05_$tag > $o/step_profile.log 2>&1
tools/pmc_step.sh $o/pmc_step.json > $o/pmc_step.log 2>&1
tools/pmc_layer.sh $o/pmc_dconv3.json --shape 16 64 32 256 512 --bn --p16 > /dev/null 2>&1
tools/pmc_layer.sh $o/pmc_dconv2_1.json --shape 16 256 128 64 128 --stride 2 --bn --p16 > /dev/null 2>&1
tools/pmc_layer.sh $o/pmc_gconv6_1.json --shape 16 256 256 32 32 --transposed --bn --p16 > /dev/null 2>&1
# the vision-infused step: kernel stats of the three-stream step, counters of ResNet layer1 / layer2 alone (forward, data gradient, weight gradient)
d=/tmp/avp_$$; rm -rf $d; mkdir -p $d
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python "$GRAFT_REPO_ROOT/bench.py" --config av --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extra > $d/log.txt 2>&1)
f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $o/av_kernel_stats.csv
tools/pmc_layer.sh $o/pmc_resnet_layer1.json --shape 1024 56 56 64 64 --bn --p16 --iters 3 > /dev/null 2>&1
tools/pmc_layer.sh $o/pmc_resnet_layer2.json --shape 1024 28 28 128 128 --bn --p16 --iters 3 > /dev/null 2>&1
tail -3 $o/pmc_step.log; python -c "
import json; d=json.load(open('$o/bench_line.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('step_floor_ms'), d['roofline'].get('frac_of_floor'))"
