L=vision-infused-audio-inpainter-viai_amd/libviai_hip.so
cp $L /tmp/lib_orig.so
for v in OLD NEW OLD NEW; do cp tools/ab/lib$v.so $L; echo "== $v"; python tools/ab/sk_check.py 2>&1 | grep -v Warn | tail -3; done
cp /tmp/lib_orig.so $L
