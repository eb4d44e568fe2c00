set -x
mkdir -p gpurun_out/r05_b
tools/pmc_layer.sh gpurun_out/r05_b/pmc_cb5_dma.json --shape 16 128 128 32 32 --transposed --bn --p16
tools/pmc_layer.sh gpurun_out/r05_b/pmc_dconv2_1.json --shape 16 256 128 64 128 --stride 2 --bn --p16
python - <<'PY'
import json
for f in ("gpurun_out/r05_b/pmc_cb5_dma.json", "gpurun_out/r05_b/pmc_dconv2_1.json"):
    d = json.load(open(f))
    for k, v in d.items():
        if "conv" in k or "wgrad" in k:
            print(k[:80])
            print("   ", {c: round(x) for c, x in v.items()})
PY
