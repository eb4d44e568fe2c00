# round-5 evidence run: bench line, kernel stats, timelines, counters.   bash tools/r05_final.sh <tag>
tag=$1; o=gpurun_out/r05_$tag; mkdir -p $o
python bench.py > $o/bench.log 2>&1; grep '^{"metric"' $o/bench.log > $o/bench_line.json
tools/step_profile.sh $o r05_$tag > $o/step_profile.log 2>&1
tools/pmc_step.sh $o/pmc_step.json > $o/pmc_step.log 2>&1
tools/pmc_layer.sh $o/pmc_dconv3.json --shape 16 64 32 256 512 --bn --p16 > /dev/null 2>&1
tools/pmc_layer.sh $o/pmc_dconv2_1.json --shape 16 256 128 64 128 --stride 2 --bn --p16 > /dev/null 2>&1
tools/pmc_layer.sh $o/pmc_gconv6_1.json --shape 16 256 256 32 32 --transposed --bn --p16 > /dev/null 2>&1
tail -3 $o/pmc_step.log; python -c "
import json; d=json.load(open('$o/bench_line.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('step_floor_ms'), d['roofline'].get('frac_of_floor'))"
