#!/bin/bash
L=vision-infused-audio-inpainter-viai_amd/libviai_hip.so
cp $L /tmp/lib_orig.so
for v in $1; do cp tools/ab/lib$v.so $L; echo "== $v"; python tools/wn_pipe_stamps.py 700 2>&1 | tail -${2:-12}; done
cp /tmp/lib_orig.so $L
