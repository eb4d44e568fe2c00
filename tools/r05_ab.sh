#!/bin/bash
# round-5 A/B: per-layer kernel times (tools/layer_stats.sh) with a switch off / on.   tools/r05_ab.sh <outdir> <ENVVAR>
out=$1; var=$2; mkdir -p "$out"
run() { tag=$1; shift; for v in 0 1; do env $var=$v tools/layer_stats.sh "${tag}_$v" "$@" 2>&1 | tee -a "$out/ab_$var.txt"; done; }
run cb5 --shape 16 128 128 32 32 --transposed --bn --p16
run c61 --shape 16 256 256 32 32 --transposed --bn --p16
run cb4 --shape 16 64 128 32 32 --transposed --bn --p16
