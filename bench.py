#!/usr/bin/env python
"""bench.py — VIAI inpainting-GAN train-step throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full G+D train step on one batch of
synthetic MUSICES-shaped masked mel-spectrograms resident in HBM:
E+G forward, 3x D forward, D backward x2, D-frozen dgrad, G+E backward, 2x Adam.
Workload = BASELINE.json configs[1]: audio-only G + PatchGAN D, 256x256 mel,
batch 16 per GPU; fp32 tensors, conv products on the 16-bit matrix cores through the
f16x2 / bf16x3 operand splits with fp32 accumulation (the `config.math` string of the
JSON line is generated from the switches in effect).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

METRIC = "masked mel-spectrogram clips/sec (G+D train step, 256x256 b16) at 1/2/4/8 GPUs"
METRIC_WAVENET = "wavenet_vocoder incremental synthesis samples/sec (24 layers / 512 ch, 16 kHz, batch 8 streams)"
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense 16-bit MFMA peak
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s measured with a float4 copy)
# conv math: f16x2 split (three fp16 MFMA products per fp32 MAC: ceiling 2500 / 3) by default, bf16x3 (six products: 2500 / 6)
# under VIAI_F16X2=0, exact fp32 MFMA (157.3) under VIAI_MATH=fp32.  The library names the family every call ran
# (viai_conv2d_last_kernel); the suffix of the name says which arithmetic, and peak_of() prices it.
BF3 = os.environ.get("VIAI_MATH", "") != "fp32"
F16X2 = BF3 and os.environ.get("VIAI_F16X2", "1") != "0"

# the kernel behind each family the library reports (names only: DESIGN.md section 3 describes them; which family a call runs is the library's decision)
FAMILY_KERNELS = {
    "wgrad_patch_f16x2": "wgrad_patch_f16_kernel<1,128,64,4,1>", "wgrad_patch_narrow_f16x2": "wgrad_patch_f16_kernel<1,32,32,4,4>",
    "wgrad_patch_s2_f16x2": "wgrad_patch_f16_kernel<2,128,32,2,1>", "wgrad_patch64_f16x2": "wgrad_patch_f16_kernel<1,64,64,2,1>",
    "wgrad_bf3_f16x2": "wgrad_bf3_kernel<2,TM,TN>", "wgrad_bf3_bf16x3": "wgrad_bf3_kernel<3,TM,TN>", "wgrad_mfma_f32": "wgrad_mfma_kernel",
    "wgrad32_all_taps_f32": "wgrad32_halo_kernel", "halo_wide256_f16x2": "conv_wide_dma_kernel<1,2>", "lin_dma_f16x2": "conv_lin_dma_kernel<PW>",
    "halo_wide128_f16x2": "conv_halo_wide_f16_kernel<1,4,4,1>", "halo_wide64_f16x2": "conv_halo_wide_f16_kernel<2,2,2,1>",
    "halo_wide32_f16x2": "conv_halo_wide_f16_kernel<4,1,1,1>", "halo_wide_s2_f16x2": "conv_wide_dma_kernel<2,1>", "halo_c32_f16x2": "conv_halo_c32_dma_kernel",
    "halo_f16x2": "conv_halo_bf3_kernel<CIN,TN,2>", "halo_bf16x3": "conv_halo_bf3_kernel<CIN,TN,3>", "dgrad_s2_patch_f16x2": "conv_dgrad_s2_patch_kernel",
    "dgrad_s2_f16x2": "conv_dgrad_s2_bf3_kernel<2>", "dgrad_s2_bf16x3": "conv_dgrad_s2_bf3_kernel<3>", "igemm128x256_f16x2": "conv_igemm_bf3_frag_kernel<2,2,2,2,4>",
    "igemm128x128_f16x2": "conv_igemm_bf3_frag_kernel<2,2,2,2,2>", "igemm128x128_bf16x3": "conv_igemm_bf3_frag_kernel<3,2,2,2,2>",
    "igemm_sk32x32_f16x2": "conv_igemm_bf3_sk_kernel<2>", "igemm_sk32x32_bf16x3": "conv_igemm_bf3_sk_kernel<3>", "stem_f16x2": "stem_fwd_f16_kernel",
    "wgrad_stem_f16x2": "stem_wgrad_f16_kernel", "direct": "cin1_* / cout1_* streaming kernels",
}
# kernel-name fragment of a family in the rocprofv3 counter summaries under profiles/ (traffic of the dominant kernel)
PMC_KERNEL_OF = {"wgrad_patch_f16x2": "wgrad_patch_f16_kernel<1, 128", "wgrad_patch_s2_f16x2": "wgrad_patch_f16_kernel<2", "wgrad_patch_narrow_f16x2": "wgrad_patch_f16_kernel<1, 32",
                 "wgrad_bf3_f16x2": "wgrad_bf3_kernel", "halo_wide256_f16x2": "conv_wide_dma_kernel<1, 2>", "halo_wide128_f16x2": "conv_halo_wide_f16_kernel", "lin_dma_f16x2": "conv_lin_dma_kernel",
                 "igemm128x256_f16x2": "frag_kernel<2, 2, 2, 2, 4", "igemm128x128_f16x2": "frag_kernel<2", "igemm128x128_bf16x3": "frag_kernel<3"}


def math_string():
    """what the conv kernels compute in, from the switches in effect (csrc/conv_api.hip reads the same environment; DESIGN.md section 3 has the long form)"""
    if not BF3:
        return "VIAI_MATH=fp32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32) in every conv kernel"
    if F16X2:
        f16b = os.environ.get("VIAI_F16_BACKWARD", "1") != "0"
        return "fp32 tensors + accumulation; f16x2 operand split on fp16 MFMA (22 significand bits, 3 products/MAC)%s" % ("" if f16b else "; backward bf16x3")
    return "fp32 tensors + accumulation; bf16x3 operand split on bf16 MFMA (6 products/MAC) (VIAI_F16X2=0)"


# ---- the contract line stays small (round 6) --------------------------------------------------------------------------------------
# Round 5's line grew to 38 KB (quoted child lines, paragraphs of prose) and the driver could not parse it.  Everything bench.py measures is written
# to a DETAIL file (--detail, default gpurun_out/bench_detail_<config>.json); the ONE stdout line is a fixed projection of it: numbers, short names,
# every string <= 120 characters, < 8 KB in total (tests/test_bench_gpu.py holds it to that).  Prose lives in DESIGN.md.
LINE_LIMIT = 8000
STR_LIMIT = 120


def _short(v):
    if isinstance(v, str):
        return v if len(v) <= STR_LIMIT else v[:STR_LIMIT - 1] + "~"
    if isinstance(v, dict):
        return {k: _short(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_short(x) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d and d[k] is not None} if isinstance(d, dict) else {}


def compact_leg(leg):
    """a child run's contract line reduced to what the parent quotes under `extra`"""
    if "error" in leg:
        return {"error": _short(leg["error"]), "leg_wall_s": leg.get("leg_wall_s")}
    out = _pick(leg, ("value", "unit", "ms_per_step", "steps", "leg_wall_s"))
    out["workload"] = _short(leg.get("config", {}).get("workload", ""))[:60]
    out["roofline"] = _pick(leg.get("roofline", {}), ("bound", "frac", "achieved", "peak", "unit", "kernel", "family", "avg_launch_us", "launches_per_step",
                                                       "step_floor_ms", "frac_of_floor", "frac_of_chain_floor"))
    if "cpu_baseline" in leg:
        out["cpu_baseline"] = _pick(leg["cpu_baseline"], ("value", "unit", "cores", "kind"))
    return _short(out)


def compact_line(full):
    """the projection of the detail record that goes to stdout"""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"))
    out["vs_baseline"] = full.get("vs_baseline")
    cfg = full.get("config", {})
    out["config"] = _pick(cfg, ("workload", "global_batch", "parallelism", "launch", "math", "host_enqueue_ms_per_step", "loss_d", "loss_g",
                                "algorithmic_gflop_per_step", "step_tflops", "comm_ms_exposed", "ranks_share_devices", "real_time_factor_16khz", "launches_per_step"))
    lm = cfg.get("launch_modes")
    if lm:
        out["config"]["launch_modes"] = {k: _pick(v, ("host_ms_per_step_idle_queue", "ms_per_step", "launches")) for k, v in lm.items()}
    rf = full.get("roofline")
    if rf:
        r = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "kernel", "family", "launches_per_step", "avg_launch_us", "algorithmic_gflop_per_launch",
                       "algorithmic_bytes_per_step", "frac_by_kernel", "conv_time_share_by_kernel", "conv_frac_whole_step", "conv_ms_per_step",
                       "conv_gflop_per_step_timed", "step_floor_ms", "frac_of_floor", "step_floor_at_peak_ms", "frac_of_floor_at_peak", "frac_of_chain_floor"))
        r["traffic"] = rf.get("traffic")
        if rf.get("traffic_source"):
            r["traffic_source"] = rf["traffic_source"]
        if "standalone" in rf:
            r["standalone"] = _pick(rf["standalone"], ("frac", "avg_launch_us", "conv_ms_per_step"))
        sf = rf.get("step_floor")
        if sf:
            r["single_stream_ms"] = sf.get("single_stream_ms")
            r["launches_per_step_all"] = sf.get("launches_per_step")
        if "power_limit" in rf:
            r["power_limit_ratio"] = rf["power_limit"].get("ratio")
        if "mfma_busy" in rf:
            r["mfma_busy"] = _pick(rf["mfma_busy"], ("conv_kernels_time_weighted", "whole_step", "dominant_kernel", "source"))
        out["roofline"] = r
    st = full.get("stages")
    if st:
        out["stages"] = _pick(st, ("stft_mel_gbps", "stft_mel_us", "mask_gbps", "mask_us"))
        if "large_batch" in st:
            out["stages"]["large_batch"] = _pick(st["large_batch"], ("clips", "stft_mel_gbps", "stft_mel_us", "stft_mel_frac_of_hbm_peak", "mask_gbps", "mask_frac_of_hbm_peak"))
    if "extra" in full:
        out["extra"] = {k: compact_leg(v) for k, v in full["extra"].items()}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample"))
        if "small_batch" in cb:
            out["cpu_baseline"]["small_batch_value"] = cb["small_batch"].get("value")
    if full.get("detail_file"):
        out["detail_file"] = full["detail_file"]
    out = _short(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:                                   # belt and braces: drop the per-family tables before anything the contract names
        for k in ("conv_time_share_by_kernel", "frac_by_kernel"):
            out.get("roofline", {}).pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) <= LINE_LIMIT, len(line)
    return line


def emit(full, args):
    """write the whole record to the detail file (never to stdout / stderr: the driver keeps only a tail) and print the contract line LAST"""
    path = args.detail
    if path is None:
        d = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_detail_%s.json" % args.config)
        except OSError:
            path = ""
    if path:
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            full["detail_file"] = os.path.relpath(path, ROOT)
        except OSError:
            pass
    sys.stderr.flush()
    print(compact_line(full), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 30; --config wavenet: 2048 synthesis time steps)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 5; wavenet: 64)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--graph", action="store_true", help="replay three captured HIP graphs per step (single stream) instead of "
                    "launching eagerly with the weight gradients on a side stream (the default, measured faster)")
    ap.add_argument("--plan", action="store_true", help="record the step once (stream capture) and replay it from C as a launch plan: "
                    "the eager step's kernels, streams and edges without the per-launch host work")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)    # kept for old command lines: eager is the default
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the bounded legs on the other configs / the exact-fp32 mode that the default N = 1 run appends under `extra`")
    ap.add_argument("--layers", action="store_true", help="print the per-layer conv timing table to stderr")
    ap.add_argument("--detail", default=None, help="file for the full record (default gpurun_out/bench_detail_<config>.json; '' = none)")
    ap.add_argument("--config", choices=["audio", "av", "av_msd", "wavenet"], default="audio",
                    help="audio = BASELINE configs[1] (the metric); av = configs[2] (+ResNet-18 visual branch, N = T/4 frames); "
                         "av_msd = configs[3] model (+3-scale D); wavenet = configs[4] (incremental synthesis, 8 streams, reference-size stack)")
    ap.add_argument("--cpu-batch", type=int, default=2, help="clips per CPU-baseline step (bounded sample)")
    ap.add_argument("--share-gpu", action="store_true", help="--gpus N on a box with fewer than N devices: the ranks share the visible device(s) and "
                    "exchange over gloo (RCCL refuses two ranks on one device).  Exercises the launch contract and the bucketed exchange; not a measurement")
    a = ap.parse_args()
    wn = a.config == "wavenet"
    a.steps = a.steps if a.steps is not None else (2048 if wn else 30)
    a.warmup = a.warmup if a.warmup is not None else (64 if wn else 5)
    return a


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: run the same command line as N ranks under torch.distributed.run (one
    process per GPU, RCCL), pass rank 0's JSON line through and exit with the launcher's status."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < args.gpus:
        if not args.share_gpu:
            sys.exit("bench.py: --gpus %d but %d device(s) visible (add --share-gpu to run the ranks on the visible device(s) over gloo: "
                     "a launch-contract check, not a measurement)" % (args.gpus, n_dev))
        env["VIAI_DIST_BACKEND"] = "gloo"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


FAMILY_CALLS = ("viai_conv2d_fwd", "viai_conv2d_fwd_amax", "viai_conv2d_dgrad", "viai_conv2d_dgrad_f16", "viai_conv2d_wgrad", "viai_conv2d_wgrad_f16",
                "viai_conv2d_cin1_bn_fwd", "viai_conv2d_cin1_bn_wgrad",
                # the same launches on pre-split (P16) operands (ABI 13)
                "viai_conv2d_fwd_p16", "viai_conv2d_dgrad_f16_p16", "viai_conv2d_wgrad_f16_p16", "viai_conv2d_cin1_bn_fwd_p16")


class KernelTimer:
    """HIP-event timing of the conv launches on the stream each one is launched on.  Wraps the ctypes entry points of
    libviai_hip.so; flops are the ALGORITHMIC 2*MACs of each call; the kernel family of a call is what the LIBRARY reports it
    launched (`viai_conv2d_last_kernel`, include/viai_hip.h) -- bench.py holds no copy of the dispatch rules."""

    def __init__(self, lib):
        self.lib = lib
        self.records = []          # (family, flops, n_launches, ev0, ev1, entry point, layer key)
        self.orig = {}
        self._buf = ctypes.create_string_buffer(64)

    @staticmethod
    def _geom(d):
        oh = (d.IH - 1 - 2 * d.ph + d.kh) if d.transposed else (d.IH + 2 * d.ph - d.kh) // d.sh + 1    # VIAI layers: no dilation
        ow = (d.IW - 1 - 2 * d.pw + d.kw) if d.transposed else (d.IW + 2 * d.pw - d.kw) // d.sw + 1
        cin = d.C1 + d.C2
        flops = 2.0 * d.N * oh * ow * d.Cout * cin * d.kh * d.kw
        return cin, flops

    def install(self):
        lib = self.lib
        from viai_amd import ops as ops_mod

        def wrap(name, counts=lambda args: True):
            fn = getattr(lib, name)
            self.orig[name] = fn

            def timed(desc_ref, *args):
                d = desc_ref._obj
                cin, flops = self._geom(d)
                if not counts(args):
                    flops = 0.0                      # a recomputation of the same conv (fused Cin = 1 layer): time yes, algorithmic flops no
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                # the events go on the stream the kernel is launched on (last argument): the weight gradients run on
                # ops.WGRAD_STREAM next to the main backward chain, and are timed there, overlap included
                side = ops_mod.WGRAD_STREAM
                strm = side if (side is not None and args and args[-1] == side.cuda_stream) else torch.cuda.current_stream()
                e0.record(strm)
                r = fn(desc_ref, *args)
                e1.record(strm)
                nl = int(lib.viai_conv2d_last_kernel(self._buf, 64))       # same thread as the call above: the tag is per thread
                fam = self._buf.value.decode() or "unknown"
                self.records.append((fam, flops, max(nl, 1), e0, e1, name, (d.N, d.IH, d.IW, d.C1 + d.C2, d.Cout, d.kh, d.kw, d.sh, d.sw, d.transposed)))
                return r
            setattr(lib, name, timed)

        for name in FAMILY_CALLS:
            # fused Cin = 1 conv + BatchNorm layer: statistics pass (z = NULL, args[7]) + apply pass: the flops count once
            wrap(name, counts=(lambda args: bool(args[7])) if name == "viai_conv2d_cin1_bn_fwd" else (lambda args: True))

    def per_layer(self):
        torch.cuda.synchronize()
        agg = {}
        for f, flops, nl, e0, e1, name, key in self.records:
            t = agg.setdefault((name, f, key), [0.0, 0.0, 0])
            t[0] += flops; t[1] += e0.elapsed_time(e1) * 1e-3; t[2] += 1
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        return ["%-18s %-26s %-44s n=%3d avg %8.1f us  %6.1f TF/s" % (k[0][5:], k[1], str(k[2]), v[2], v[1] / v[2] * 1e6, v[0] / v[1] * 1e-12)
                for k, v in rows]

    def uninstall(self):
        for k, fn in self.orig.items():
            setattr(self.lib, k, fn)

    def summary(self):
        torch.cuda.synchronize()
        fam = {}
        for f, flops, nl, e0, e1, _n, _k in self.records:
            t = fam.setdefault(f, [0.0, 0.0, 0])
            t[0] += flops
            t[1] += e0.elapsed_time(e1) * 1e-3
            t[2] += nl
        return fam


def peak_of(family):
    """ceiling of a kernel family's arithmetic, from the suffix the library puts on the name"""
    if family.endswith("_f16x2"):
        return MFMA_BF16_PEAK_TFLOPS / 3.0
    if family.endswith("_bf16x3"):
        return MFMA_BF16_PEAK_TFLOPS / 6.0
    return MFMA_F32_PEAK_TFLOPS


# ---- whole-step floor (round 5) ---------------------------------------------------------------------------------------------------
# Per launch: floor_us = max(algorithmic flops / the SUSTAINED ceiling of its arithmetic, algorithmic bytes / the measured copy bandwidth).
SUSTAINED_TFLOPS = {"f16x2": 1677.0 / 3.0,        # v_mfma_f32_32x32x16_f16 on random operands at the clock the power budget leaves (profiles/r04_d_mfma_shapes.txt), 3 products per MAC
                    "bf16x3": 1810.0 / 6.0,       # bf16 MFMA, random operands (profiles/r01_e_mfma_probe.txt), 6 products per MAC
                    "f32": 155.0}                 # v_mfma_f32_32x32x2_f32 micro-benchmark ceiling = the fp32 vector rate (MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6300.0                            # float4 copy, MI355X_MICROARCH.md


def _arith_of(family):
    return "f16x2" if family.endswith("_f16x2") else "bf16x3" if family.endswith("_bf16x3") else "f32"


def _conv_bytes(d):
    oh = (d.IH - 1 - 2 * d.ph + d.kh) if d.transposed else (d.IH + 2 * d.ph - d.kh) // d.sh + 1
    ow = (d.IW - 1 - 2 * d.pw + d.kw) if d.transposed else (d.IW + 2 * d.pw - d.kw) // d.sw + 1
    cin = d.C1 + d.C2
    return 4.0 * (d.N * d.IH * d.IW * cin + d.N * oh * ow * d.Cout + d.Cout * cin * d.kh * d.kw), d.N * oh * ow


# algorithmic bytes of the non-conv entry points of the step, from their scalar arguments (tools/step_calls.py prints the argument lists
# this table was written against); M = pixels, C = channels, fp32 elements.  Entry points not listed count 0 bytes (their time stays in
# single_stream_ms, their floor is 0: the floor errs low).
NONCONV_BYTES = {
    "viai_bn_act_fwd_p16": lambda a: 8.0 * a[7] * a[8],                                   # read y, write z
    "viai_bn_act_fwd_amax": lambda a: 8.0 * a[4] * a[5],
    "viai_bn_act_fwd": lambda a: 8.0 * a[4] * a[5],
    "viai_bn_act_bilinear_fwd_p16": lambda a: 4.0 * a[7] * a[12] * (a[8] * a[9] + a[10] * a[11]),      # read y at (IH, IW), write z at (OH, OW)
    "viai_bn_act_bilinear_fwd_amax": lambda a: 4.0 * a[4] * a[9] * (a[5] * a[6] + a[7] * a[8]),
    "viai_bn_act_bwd_p16": lambda a: 20.0 * a[11] * a[12],                                # reduce: dz, y; apply: dz, y -> dy
    "viai_bn_act_bwd_amax": lambda a: 20.0 * a[11] * a[12],
    "viai_bilinear_ac_bwd": lambda a: 4.0 * a[2] * a[7] * (a[3] * a[4] + a[5] * a[6]),
    "viai_adam_step": lambda a: 28.0 * a[4],                                              # p, g, m, v read; p, m, v written
    "viai_bce_fwd": lambda a: 8.0 * a[2], "viai_bce_bwd": lambda a: 12.0 * a[2],
    "viai_l1_fwd": lambda a: 8.0 * a[2], "viai_l1_bwd": lambda a: 12.0 * a[2],
    "viai_act_bwd_from_output": lambda a: 12.0 * a[3],
    "viai_colsum": lambda a: 4.0 * a[1] * a[2],
}
# entry points that take a viai_conv2d descriptor first and stream one multi-channel tensor: bytes per (pixel x channel) of that tensor
DESC_STREAM_BYTES = {
    "viai_conv2d_cin1_bn_fwd": lambda a: 4.0 if a[6] else 0.0,                            # statistics pass writes nothing; the apply pass writes z
    "viai_conv2d_cin1_bn_fwd_p16": lambda a: 4.0,
    "viai_conv2d_cin1_bn_bwd": lambda a: 4.0, "viai_conv2d_cin1_bn_wgrad": lambda a: 4.0, "viai_conv2d_cin1_bn_dgrad": lambda a: 4.0,      # read dz once each
    "viai_pair_cout1_fwd_dots": lambda a: 4.0, "viai_pair_cout1_fwd": lambda a: 8.0, "viai_pair_cout1_wgrad": lambda a: 4.0,
    "viai_pair_cout1_bn_bwd_p16": lambda a: 20.0, "viai_pair_cout1_bn_bwd": lambda a: 20.0,
}


class StepFloor:
    """Every launch of ONE single-stream step (weight gradients and D(real) back on the main stream): HIP events around every library call,
    the call's algorithmic (flops, bytes), floor = max(flops / sustained ceiling, bytes / copy bandwidth)."""

    def __init__(self, lib):
        from viai_amd import _lib as L
        # entry points that launch: their last parameter is the stream (void*); geometry / size queries are left alone
        self.lib = lib
        self.names = [n for n, (_r, at) in L.SIGNATURES.items() if hasattr(lib, n) and at and at[-1] is ctypes.c_void_p and not n.endswith(("_ok", "_blocks"))]
        self.rec, self.orig = [], {}
        self._buf = ctypes.create_string_buffer(64)

    def install(self):
        lib = self.lib
        for name in self.names:
            fn = getattr(lib, name)
            self.orig[name] = fn
            setattr(lib, name, self._wrap(name, fn))

    def _wrap(self, name, fn):
        lib = self.lib
        is_conv = name in FAMILY_CALLS

        def timed(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*args)
            e1.record()
            flops, nbytes, fam, key = 0.0, 0.0, "", ()
            d = getattr(args[0], "_obj", None) if args else None
            if is_conv and d is not None:
                lib.viai_conv2d_last_kernel(self._buf, 64)
                fam = self._buf.value.decode() or "unknown"
                flops = KernelTimer._geom(d)[1]
                if name == "viai_conv2d_cin1_bn_fwd" and not args[7]:
                    flops = 0.0
                nbytes = _conv_bytes(d)[0]
                key = (d.N, d.IH, d.IW, d.C1 + d.C2, d.Cout, d.kh, d.kw, d.sh, d.sw)
                if name in DESC_STREAM_BYTES:                      # the fused Cin = 1 layer: its input is one channel, the traffic is the multi-channel side
                    nbytes = DESC_STREAM_BYTES[name](args) * _conv_bytes(d)[1] * d.Cout
            elif d is not None and name in DESC_STREAM_BYTES:
                px = _conv_bytes(d)[1] if (d.C1 + d.C2) == 1 else d.N * d.IH * d.IW
                ch = d.Cout if (d.C1 + d.C2) == 1 else d.C1 + d.C2
                nbytes = DESC_STREAM_BYTES[name](args) * px * ch
                key = (d.N, d.IH, d.IW, d.C1 + d.C2, d.Cout)
            elif name in NONCONV_BYTES:
                nbytes = float(NONCONV_BYTES[name](args))
                key = tuple(x for x in args if isinstance(x, int) and 0 < x < (1 << 31))[:6]
            self.rec.append((name, fam, key, flops, nbytes, e0, e1))
            return r
        return timed

    def uninstall(self):
        for k, fn in self.orig.items():
            setattr(self.lib, k, fn)

    def summary(self, nsteps, ms_per_step):
        torch.cuda.synchronize()
        agg = {}
        for name, fam, key, flops, nbytes, e0, e1 in self.rec:
            t = e0.elapsed_time(e1) * 1e3                            # us
            if t <= 0.0:
                continue                                             # host-only entry points (geometry queries) launch nothing
            arith = _arith_of(fam) if fam and fam != "direct" else "f32"
            f_us = flops / (SUSTAINED_TFLOPS[arith] * 1e12) * 1e6
            b_us = nbytes / (HBM_COPY_GBPS * 1e9) * 1e6
            a = agg.setdefault((name, fam, key), [0, 0.0, 0.0, 0.0, 0.0, "mfma" if f_us >= b_us else "hbm", 0.0])
            a[0] += 1; a[1] += t; a[2] += max(f_us, b_us); a[3] += flops; a[4] += nbytes
            # the same floor against the guide's PEAKS (dense 16-bit MFMA 2500 / products per MAC, fp32 MFMA 157.3, HBM 8 TB/s) instead of the sustained ceilings
            a[6] += max(flops / (peak_of(fam if fam and fam != "direct" else "f32") * 1e12) * 1e6, nbytes / (HBM_PEAK_GBPS * 1e9) * 1e6)
        tot_t = sum(a[1] for a in agg.values()) / nsteps
        tot_f = sum(a[2] for a in agg.values()) / nsteps
        tot_fp = sum(a[6] for a in agg.values()) / nsteps
        by_bound = {"mfma": 0.0, "hbm": 0.0}
        for a in agg.values():
            by_bound[a[5]] += a[2] / nsteps
        fam_hbm = {}
        for (name, fam, key), a in agg.items():
            if fam:
                h = fam_hbm.setdefault(fam, [0.0, 0.0, 0.0])
                h[0] += a[1]; h[1] += a[3] / (SUSTAINED_TFLOPS[_arith_of(fam) if fam != "direct" else "f32"] * 1e12) * 1e6; h[2] += a[4] / (HBM_COPY_GBPS * 1e9) * 1e6
        gaps = sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:10]
        return {
            "step_floor_ms": round(tot_f * 1e-3, 3),
            "frac_of_floor": round(tot_f * 1e-3 / ms_per_step, 4),
            "step_floor_at_peak_ms": round(tot_fp * 1e-3, 3),
            "frac_of_floor_at_peak": round(tot_fp * 1e-3 / ms_per_step, 4),
            "single_stream_ms": round(tot_t * 1e-3, 3),
            "floor_ms_by_bound": {k: round(v * 1e-3, 3) for k, v in by_bound.items()},
            "launches_per_step": round(sum(a[0] for a in agg.values()) / nsteps),
            "family_bound": {f: {"bound": "hbm" if h[2] > h[1] else "mfma", "frac_of_own_floor": round(max(h[1], h[2]) / h[0], 3),
                                 "hbm_frac_of_copy_rate": round(h[2] / h[0], 3)} for f, h in sorted(fam_hbm.items()) if h[0] > 0},
            "largest_gaps": [{"call": k[0][5:], "family": k[1], "shape": list(k[2]), "launches_per_step": round(a[0] / nsteps, 2),
                              "measured_us": round(a[1] / a[0], 1), "floor_us": round(a[2] / a[0], 1), "bound": a[5],
                              "gap_us_per_step": round((a[1] - a[2]) / nsteps, 1)} for k, a in gaps],
            "model": ("per launch of one single-stream step (HIP events around every libviai_hip.so call, %d steps): floor = max(2 x MACs / sustained "
                      "ceiling of the launch's arithmetic [f16x2 %.0f, bf16x3 %.0f, fp32 %.0f TFLOP/s: random-operand MFMA rates under the power limit], algorithmic "
                      "bytes / %.1f TB/s [float4 copy]); conv bytes = input + output + weights once, BatchNorm forward 8 B / element, backward 20 B / element "
                      "(two passes), resize in + out, Adam 28 B / parameter; frac_of_floor = step_floor_ms / ms_per_step of the timed (three-stream) step"
                      % (nsteps, SUSTAINED_TFLOPS["f16x2"], SUSTAINED_TFLOPS["bf16x3"], SUSTAINED_TFLOPS["f32"], HBM_COPY_GBPS * 1e-3)),
        }


def cpu_baseline(args):
    """The oracle's train step on the host cores, same step and SAME workload as the GPU line: 16 clips of 256 x 256 per step
    (SURVEY.md section 8d), all cores the process may run on handed to torch (`cores`), best of the steps that fit ~20 s.
    A 2-clip step at 16 threads -- where torch's CPU convolutions are most efficient per clip -- is reported beside it."""
    from oracle import viai_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count()

    def one_step(b, f, t, tag):
        s = O.cf_uniform(tag + ".s", (b, 1, f, t))
        mask = O.make_mask(b, t, tag + ".mask")
        E, G, D = O.encoder_state(), O.decoder_state(), O.disc_state()
        oG, oD = O.new_optimizers(E, G, D)
        t0 = time.perf_counter()
        O.train_step(E, G, D, oG, oD, s, mask)
        return time.perf_counter() - t0

    def sample(b, threads, budget_s, max_steps):
        torch.set_num_threads(threads)
        one_step(1, 80, 32, "bench.cpu.warm")                   # library warm-up at a tiny shape (untimed)
        times = [one_step(b, args.bins, args.frames, "bench.cpu")]
        while sum(times) < budget_s and len(times) < max_steps:
            times.append(one_step(b, args.bins, args.frames, "bench.cpu"))
        return min(times), len(times), sum(times)
    # torch's CPU convolutions stop scaling at ~16 threads and collapse under oversubscription: one 16-clip step measured 5.8 s at 16
    # threads, 6.4 s at 32, 9.7 s at 64 and 192 s at 256 on the 256-logical-CPU host of an MI355X box -- so 16 is the best case
    full_threads = max(1, min(avail, 16))
    dt, n, tot = sample(args.batch, full_threads, 15.0, 4)
    out = {"value": round(args.batch / dt, 4), "unit": "clips/s", "cores": full_threads, "kind": "port",
           "sample": "oracle train_step, torch CPU fp32, %d of %d CPUs, %d clips %dx%d/step, best of %d (%.1f s), %.2f s/step"
                     % (full_threads, avail, args.batch, args.bins, args.frames, n, tot, dt)}
    t2 = max(1, min(avail, 16))
    dt2, n2, tot2 = sample(args.cpu_batch, t2, 6.0, 64)
    out["small_batch"] = {"value": round(args.cpu_batch / dt2, 4), "unit": "clips/s", "cores": t2,
                          "sample": "%d clips per step at %d threads, best of %d step(s) (%.1f s)" % (args.cpu_batch, t2, n2, tot2)}
    return out


def launch_modes(hp, dev, s, mask, steps=30):
    """host time per step.  The loop figure `host_enqueue_ms_per_step` is back-pressure-bound (the host
    runs ahead until the device queues are full, then waits inside a launch), so the host cost is measured separately: time to
    enqueue ONE step into an idle queue, for the eager step (Python -> ctypes -> launch) and for the launch plan (csrc/plan.hip:
    the same launches, streams and edges replayed from C; bitwise the eager step, tests/test_networks_gpu.py::
    test_launch_plan_replays_the_eager_step_bitwise), plus each mode's steady-state ms/step in this run."""
    import statistics
    from viai_amd.model import AudioModel
    out = {}
    for name, kw in (("eager", {}), ("plan", {"use_plan": True})):
        m = AudioModel(hp, device=dev, **kw)
        m.set_inputs(s, mask)
        for i in range(8):
            m.optimize_parameters(i)
        torch.cuda.synchronize()
        host = []
        for i in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.optimize_parameters(8 + i)
            host.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            m.optimize_parameters(20 + i)
        torch.cuda.synchronize()
        out[name] = {"host_ms_per_step_idle_queue": round(statistics.median(host), 3),
                     "ms_per_step": round((time.perf_counter() - t0) * 1e3 / steps, 3)}
        if name == "plan":
            info = m.plan_info()
            out[name]["launches"] = sum(v[0] for v in info)
            out[name]["cross_stream_events"] = sum(v[6] for v in info)
        del m
    return out


def dvfs_probe(dev):
    """The dominant kernel's launch on random and on all-zero operands (same instruction stream, same cycles): the ratio is the clock the power budget
    takes away under this kernel (MI355X_MICROARCH.md 'DVFS give-back'; DESIGN.md 9.5b, profiles/r04_d_clock_dconv3*.txt)."""
    from viai_amd import ops
    N, H, W, Ci, Co = 16, 64, 32, 256, 512                        # D.conv3 of the metric config (Discriminator_Networks.py:44)
    res = {}
    for name, k in (("random", 1.0), ("zero", 0.0)):
        x = (torch.rand(N, H, W, Ci, device=dev) * 2 - 1) * k
        w = (torch.rand(Co, Ci, 3, 3, device=dev) - 0.5) * 0.1 * k
        x._viai_amax = torch.ones(1, device=dev)                  # the operand magnitude a producer would publish (|x| <= 1): no abs-max pass in the timed loop
        with torch.no_grad():
            for _ in range(5):
                ops.begin_step(dev)
                ops.conv_bn_act(x, w, None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=ops.ACT_NONE)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                ops.conv_bn_act(x, w, None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=ops.ACT_NONE)
            e1.record()
            torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    return {"layer": "D.conv3 forward, 16x64x32x256 -> 512 (the dominant kernel's largest launch; here on an fp32 tensor, i.e. with the operand split inside the kernel), 20 back-to-back launches each",
            "random_operand_us": round(res["random"], 1), "zero_operand_us": round(res["zero"], 1), "ratio": round(res["zero"] / res["random"], 3),
            "note": "identical instruction stream and cycle count on all-zero operands: the difference is the shader clock under the kernel's power draw "
                    "(profiles/r04_d_clock_dconv3.txt: 1.5 GHz with random operands against 2.15 GHz with zeros, 76 % of the MFMA issue slots taken in "
                    "both) -- roofline.frac prices the launch against the 2.4 GHz peak"}


def front_end_stages(dev, batch, bins, frames):
    """north_star: achieved HBM GB/s of the STFT / mask stages.  Algorithmic bytes: STFT -> mel reads the
    waveform (4 B x 65 536 samples per clip) and writes the mel (4 B x F x T); the mask multiply reads and writes the mel once."""
    from viai_amd import ops, synth
    from viai_amd.audio import AudioConfig, MelFrontEnd

    class Cfg(AudioConfig):
        num_mels = bins
    n_samples = 65536
    fe = MelFrontEnd(Cfg)
    wav = synth.waveform(batch, n_samples).to(dev)
    mel = synth.mel_batch(batch, bins, frames, "bench.stage.s", 0).to(dev)
    mask = synth.time_mask(batch, frames, "bench.stage.mask", 0).to(dev)

    def timed(fn, n=50):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n
    with torch.no_grad():
        t_fe = timed(lambda: fe(wav))
        t_mk = timed(lambda: ops.mask_mul(mel, mask))
    fr = fe.num_frames(n_samples)
    b_fe = batch * (n_samples * 4 + bins * fr * 4)
    b_mk = 2 * 4 * batch * bins * frames
    # the same two kernels on a batch whose traffic exceeds the 256 MB Infinity Cache (1024 clips: 268 MB of waveform in, 272 MB of mel
    # out; mask: 1024 mels = 268 MB in + 268 MB out): a sustained-bandwidth figure, which the 16-clip launch (8.4 MB) cannot be
    big = 1024
    wav_b = synth.waveform(16, n_samples).to(dev).repeat(big // 16, 1)
    mel_b = torch.rand(big, 1, bins, frames, device=dev)
    mask_b = synth.time_mask(16, frames, "bench.stage.mask", 0).to(dev).repeat(big // 16, 1, 1, 1)
    with torch.no_grad():
        t_fe_b = timed(lambda: fe(wav_b), n=10)
        t_mk_b = timed(lambda: ops.mask_mul(mel_b, mask_b), n=10)
    b_fe_b, b_mk_b = big * (n_samples * 4 + bins * fr * 4), 2 * 4 * big * bins * frames
    del wav_b, mel_b, mask_b
    large = {"clips": big, "stft_mel_gbps": round(b_fe_b / t_fe_b * 1e-9, 1), "stft_mel_us": round(t_fe_b * 1e6, 1), "stft_mel_frac_of_hbm_peak": round(b_fe_b / t_fe_b / 8e12, 4),
             "mask_gbps": round(b_mk_b / t_mk_b * 1e-9, 1), "mask_us": round(t_mk_b * 1e6, 1), "mask_frac_of_hbm_peak": round(b_mk_b / t_mk_b / 8e12, 4),
             "note": "1024 clips per launch: algorithmic traffic 540 / 537 MB, beyond the 256 MB Infinity Cache -- the sustained-bandwidth figure of the two stages; "
                     "in the train step the mask is applied inside E.conv1's loads (no mask kernel) and the mel comes from the loader"}
    return {"large_batch": large,"stft_mel_gbps": round(b_fe / t_fe * 1e-9, 1), "stft_mel_us": round(t_fe * 1e6, 1), "stft_mel_frac_of_hbm_peak": round(b_fe / t_fe / 8e12, 4),
            "mask_gbps": round(b_mk / t_mk * 1e-9, 1), "mask_us": round(t_mk * 1e6, 1), "mask_frac_of_hbm_peak": round(b_mk / t_mk / 8e12, 4),
            "note": "HIP events around 50 back-to-back launches, inputs resident in HBM; algorithmic bytes: STFT->mel = waveform in (%d samples/clip) + mel out "
                    "(%d x %d), mask = mel in + out; both stages are far smaller than the GPU's caches at this batch, so the figure is launch-/latency-bound, "
                    "not a sustained-bandwidth measurement" % (n_samples, bins, fr)}


def cpu_baseline_av(args, num_D):
    """configs[2] / [3] on the host cores: the oracle's vision-infused step (oracle/viai_oracle.av_step_no_update: the reference's
    modules composed as declared) on ONE clip of the GPU line's shape -- 64 + 64 frames through two ResNet-18s forward and backward
    is ~1.4 TFLOP, the bounded sample the bench contract asks for; no Adam (the oracle's AV step has none), which favours the CPU."""
    from oracle import viai_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count()
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    nf = args.frames // 4
    s = O.cf_uniform("bench.cpu.av.s", (1, 1, args.bins, args.frames))
    mask = O.make_mask(1, args.frames, "bench.cpu.av.mask")
    video = O.cf_uniform("bench.cpu.av.video", (1, nf, 3, 224, 224), -1, 1)
    flow = O.cf_uniform("bench.cpu.av.flow", (1, nf, 2, 224, 224), -1, 1)
    E, G = O.encoder_state(), O.decoder_variant_state("image")
    D = O.msd_state(num_D) if num_D > 1 else O.disc_state()
    V = O.image_embedding2_state()
    t0 = time.perf_counter()
    O.av_step_no_update(E, G, D, V, s, mask, video, flow, num_D=num_D, lambda_contrast=0.1)
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "oracle av_step_no_update, torch CPU fp32, %d of %d CPUs, ONE clip %dx%d + %d+%d frames, fwd+bwd, no Adam, 1 step (%.1f s)"
                      % (threads, avail, args.bins, args.frames, nf, nf, dt)}


def main_wavenet(args):
    """BASELINE configs[4]: wavenet_vocoder incremental synthesis (wavenet.py:237-364) at the reference's size -- 24 layers / 4 stacks,
    512 residual + gate / 256 skip channels, 24.7 M parameters -- on 8 streams.  A step = one synthesis time step (8 samples).
    Replicas only (independent streams, sequential in T): N ranks would run N copies, so this configuration is single-GPU."""
    from viai_amd.wavenet import WaveNet
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(1234)
    B, hop = 8, 256
    T = -(-(args.warmup + args.steps) // hop) * hop                 # whole conditioning frames
    net = WaveNet(dropout=0.0).to(dev).eval()
    c = torch.rand(B, 80, T // hop, device=dev)
    timing = {"warmup": T - args.steps}
    net.incremental_forward(None, c=c, T=T, log_scale_min=-7.0, timing=timing)
    ms = timing["ms"] / timing["steps"]
    n_param = sum(p.numel() for p in net.parameters())
    fused = os.environ.get("VIAI_WN_FUSED", "1") != "0"
    pipe = timing.get("form") == "pipe"           # round 6: one persistent launch, the 27 stages own compute units and work on different streams (csrc/wavenet_pipe.hip)
    head_rows = True
    n_stage = len(net.conv_layers) + 3
    n_launch = n_stage if fused else 2 * len(net.conv_layers) + 2
    kernel_chain = ("wn_pipe_kernel: %d resident stages (10 CUs per layer, 4 + 4 + 1 for the head), weights in registers / LDS, 8-byte granule tokens" % n_stage if pipe else
                    "wn_stage_kernel x %d / wn_head_rows_kernel x 2 / wn_head_sample_kernel" % len(net.conv_layers) if fused else "wn_gate_kernel / wn_out_kernel / wn_head_kernel")
    launch_form = ("one persistent launch per 1024 time steps: a weight-stationary pipeline, up to %d stages busy at once" % B if pipe else
                   "fused stages: one dependent launch per layer" if fused else "two dependent launches per layer (VIAI_WN_FUSED=0)")
    # weight bytes one time step must stream: every layer's linearised dilated conv (3 x 512 x 512), conditioning (512 x 80), out and
    # skip 1x1s, first conv, head -- the fp32 weights the step kernels read (the up-sampling net runs once, outside the loop)
    wbytes = 4 * sum(p.numel() for n, p in net.named_parameters() if n.endswith("weight_v") and not n.startswith("upsample_conv"))
    ach = wbytes / (ms * 1e-3) * 1e-9
    out = {
        "metric": METRIC_WAVENET, "value": round(B * 1e3 / ms, 1), "unit": "samples/s", "n_gpus": 1, "steps": timing["steps"], "warmup": timing["warmup"],
        "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[4] (not the headline): wavenet_vocoder incremental synthesis, %d layers, %.1f M params, %d streams; step = 1 time step"
                               % (len(net.conv_layers), n_param * 1e-6, B),
                   "global_batch": B, "parallelism": "replicas only (independent streams)", "real_time_factor_16khz": round(1e3 / ms / 16000.0, 3),
                   "launches_per_step": (round(1.0 / 1024, 5) if pipe else n_launch), "dependent_stages_per_step": n_launch,
                   "launch": "viai_wn_pipe_run: one persistent launch" if pipe else "viai_wavenet_synth_run: C loop over the time steps", "launch_form": launch_form},
        "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
                     "kernel": ("wn_pipe_kernel (csrc/wavenet_pipe.hip): %d pipelined stages, %d streams in flight" % (n_stage, B)) if pipe else
                               "wn_stage_kernel x %d + head (csrc/wavenet.hip): %d dependent launches per time step" % (len(net.conv_layers), n_launch),
                     "kernel_chain": kernel_chain, "algorithmic_bytes_per_step": wbytes, "frac_of_chain_floor": round(n_launch * 1.7e-3 / ms, 4),
                     # the second floor: a time step is a CHAIN of n_launch dependent stages; each hand-off costs a kernel boundary (1.5 - 1.9 us
                     # between real kernels, MI355X_MICROARCH.md price list "boundary") or, inside one persistent kernel, an XCD-hierarchical grid
                     # barrier (4 - 6 us, "barrier-xcd") -- the cheaper of the two, per dependency
                     "dependency_chain_floor": {"launches_per_step": n_launch, "us_per_handoff": 1.7, "floor_us_per_step": round(n_launch * 1.7, 1),
                                                "floor_samples_per_s": round(B / (n_launch * 1.7e-6), 0),
                                                "frac_of_chain_floor": round(n_launch * 1.7e-3 / ms, 4),
                                                "note": "a stage of ONE stream cannot start before its predecessor's output is visible to it: %d x 1.7 us of hand-offs per time step "
                                                        "whatever runs them; the measured %.1f us per step = %.1f us per stage" % (n_launch, ms * 1e3, ms * 1e3 / n_launch)},
                     "note": "weight-streaming bound (SURVEY.md section 8d: incremental lower bound per time step = weight bytes / bandwidth): every time step reads "
                             "all %.1f MB of fp32 weights once, whatever level of the hierarchy serves them (they fit the 256 MB Infinity Cache, not the 32 MB of L2); "
                             "the floor at the HBM peak is %.1f us per step, the chain of dependent launches measures %.1f us" % (wbytes * 1e-6, wbytes / HBM_PEAK_GBPS * 1e-3, ms * 1e3)},
    }
    if pipe:
        # memory-side traffic of the persistent kernel: the committed counter pass (rocprofv3 cannot run inside this process)
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_wn_pipe.json")))
        if pm:
            for k, v in json.load(open(pm[-1])).items():
                if "wn_pipe" in k and "hbm_bytes_per_time_step" in v:
                    out["roofline"]["traffic"] = round(v["hbm_bytes_per_time_step"])
                    out["roofline"]["traffic_source"] = "committed: profiles/%s (rocprofv3 --pmc, 2*FETCH_SIZE+WRITE_SIZE per time step)" % os.path.basename(pm[-1])
    if not args.no_cpu_baseline:
        from oracle import wavenet_oracle as W
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count()
        threads = max(1, min(avail, 16))
        torch.set_num_threads(threads)
        cfg = W.WNConfigFull
        sd = W.wavenet_state(cfg)
        Tc = hop
        cc = torch.rand(B, cfg.cin_channels, 1)
        u1, u2 = torch.rand(B, Tc, 10).clamp(1e-5, 1 - 1e-5), torch.rand(B, Tc).clamp(1e-5, 1 - 1e-5)
        t0 = time.perf_counter()
        W.incremental_forward_ring(sd, cc, Tc, u1, u2, cfg)          # one whole conditioning frame: 256 time steps x 8 streams
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(B * Tc / dt, 1), "unit": "samples/s", "cores": threads, "kind": "port",
                               "sample": "oracle incremental_forward_ring (conv.py:17-46), torch CPU fp32, %d threads, %d streams x %d time steps (%.1f s)" % (threads, B, Tc, dt)}
    emit(out, args)


def extra_legs(budget_s=170.0):
    """The other BASELINE configs and the exact-fp32 companion as bounded child runs of this script AFTER the timed region (default N = 1 run only):
    configs[2] (vision-infused, 3 steps + a one-clip CPU step), configs[4] (WaveNet synthesis, 2048 time steps + 256 CPU time steps), configs[1] under
    VIAI_MATH=fp32 (5 steps).  Each child prints its own contract line and writes its own detail file; the parent quotes a few numbers of each
    (compact_leg).  The legs share one wall-time budget: a leg that would start after it is skipped and says so."""
    legs = {"av": (["--config", "av", "--steps", "3", "--warmup", "1"], {}),
            "wavenet": (["--config", "wavenet", "--steps", "2048", "--warmup", "64"], {}),
            "audio_exact_fp32": (["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extra"], {"VIAI_MATH": "fp32"})}
    res = {}
    t_all = time.perf_counter()
    for tag, (argv, envx) in legs.items():
        env = dict(os.environ)
        env.update(envx)
        t0 = time.perf_counter()
        left = budget_s - (t0 - t_all)
        if left < 20.0:
            res[tag] = {"error": "skipped: the legs' shared wall-time budget (%.0f s) is spent" % budget_s, "leg_wall_s": 0.0}
            continue
        try:
            detail = os.path.join(ROOT, "gpurun_out", "bench_detail_extra_%s.json" % tag)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--detail", detail] + argv, env=env, capture_output=True, text=True,
                               timeout=min(120.0, left))
            line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
            res[tag] = json.loads(line[-1]) if line else {"error": (p.stderr or p.stdout)[-400:]}
        except Exception as e:                                     # a leg that fails must not take the headline line with it
            res[tag] = {"error": repr(e)[:400]}
        res[tag]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
    return res


def shutdown(world):
    """leave the process group the clean way: every collective of this process has completed (device sync + barrier) before the
    communicator is destroyed, and nothing of torch.distributed is left for interpreter exit to tear down in an arbitrary order."""
    if world <= 1 or not torch.distributed.is_initialized():
        return
    torch.cuda.synchronize()
    torch.distributed.barrier()
    torch.cuda.synchronize()
    sys.stdout.flush()
    sys.stderr.flush()
    torch.distributed.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        spawn_ranks(args)                                    # does not return
    if args.config == "wavenet":
        if int(os.environ.get("RANK", "0")) == 0:
            main_wavenet(args)
        return
    from viai_amd import _lib, ddp, synth
    from viai_amd.model import AudioModel, StepConfig

    rank, local, world = ddp.init_from_env()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    n_dev = torch.cuda.device_count()
    shared = world > n_dev                                # --share-gpu: several ranks per device (gloo)
    local = local % n_dev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _lib.load()
    av = args.config != "audio"

    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths, hp.batch_size = args.bins, args.frames, args.batch
    torch.manual_seed(1234)                               # identical init on every rank (DDP semantics)
    if av:
        hp.use_video, hp.lambda_contrast = True, 0.1
        hp.num_D = 3 if args.config == "av_msd" else 1
    model = AudioModel(hp, device=dev, use_graph=args.graph, use_plan=args.plan)
    ddp.broadcast_arena(model.arena_G.flat)
    ddp.broadcast_arena(model.arena_D.flat)

    s = synth.mel_batch(args.batch, args.bins, args.frames, "bench.s", rank).to(dev)
    mask = synth.time_mask(args.batch, args.frames, "bench.mask", rank).to(dev)
    inputs = {}
    if av:
        nf = args.frames // 4
        inputs = {"video": synth.uniform("bench.video.r%d" % rank, (args.batch, nf, 3, 224, 224), -1, 1).to(dev),
                  "flow": synth.uniform("bench.flow.r%d" % rank, (args.batch, nf, 2, 224, 224), -1, 1).to(dev)}
    model.set_inputs(s, mask, **inputs)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        model.optimize_parameters(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.optimize_parameters(args.warmup + i)
    enqueue_ms = (time.perf_counter() - t0) * 1e3          # host time to launch the K steps (the GPU may still be running)
    barrier()
    elapsed_ms = (time.perf_counter() - t0) * 1e3
    elapsed_ms = ddp.barrier_max_ms(elapsed_ms, dev)
    ms_per_step = elapsed_ms / args.steps
    value = world * args.batch * args.steps / (elapsed_ms * 1e-3)
    losses = model.get_loss_items()
    comm_exposed = None
    if world > 1:
        # the same K steps without the collectives (replicas diverge from here on; nothing is measured afterwards):
        # what the exchange adds to the step after overlapping it with the backward / the next step's D(real) branch
        model._skip_exchange = True
        for i in range(2):
            model.optimize_parameters(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            model.optimize_parameters(i)
        barrier()
        noex_ms = ddp.barrier_max_ms((time.perf_counter() - t1) * 1e3, dev) / args.steps
        comm_exposed = round(ms_per_step - noex_ms, 3)

    out = {
        "metric": METRIC, "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("configs[1]: audio-only G (MelEncoder+MelDecoder) + PatchGAN D train step, %dx%d mel, batch %d/GPU" % (args.bins, args.frames, args.batch))
                   if not av else
                   ("configs[%d] (not the metric config): vision-infused G (2x ResNet-18, %d frames/clip) + %s, %dx%d mel, batch %d/GPU"
                    % (2 if args.config == "av" else 3, args.frames // 4, "PatchGAN D" if args.config == "av" else "3-scale D", args.bins, args.frames, args.batch)),
                   "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                   "launch": ("launch plan replayed from C (3 segments)" if args.plan else "hipGraph replay (3 segments)" if args.graph else
                              ("eager; RGB and flow ResNets as two chains on two streams" if (av and model._wgrad_stream is None)
                               else "eager, weight gradients on a side stream")),
                   "math": math_string(),
                   "host_enqueue_ms_per_step": round(enqueue_ms / args.steps, 3),
                   "loss_d": round(losses[0], 5), "loss_g": round(losses[1], 5)},
    }
    if not av and (args.batch, args.bins, args.frames) == (16, 256, 256):
        out["config"]["algorithmic_gflop_per_step"] = 1208.0
        out["config"]["step_tflops"] = round(1208.0 * 1e-3 / (ms_per_step * 1e-3), 2)
    if shared:
        out["config"]["ranks_share_devices"] = ("%d ranks on %d device(s), exchange staged through the host over gloo: a check of the launch contract and the "
                                                "bucketed exchange, NOT a scaling measurement" % (world, n_dev))
    if comm_exposed is not None:
        out["config"]["comm_ms_exposed"] = comm_exposed
        out["config"]["exchange"] = ("%s all-reduce of the flat gradient arenas in buckets launched from gradient-ready hooks inside the backward "
                                     "(D: conv3..conv4 early, rest at the end; G: two decoder buckets early, encoder at the end); the G exchange + "
                                     "Adam(E,G) + weight re-pack overlap the next step's D(real) forward/backward; comm_ms_exposed = ms_per_step - "
                                     "ms_per_step of the same steps without the collectives" % ("gloo (host-staged)" if shared else "RCCL"))

    if rank == 0 and not args.no_roofline:
        # instrumented eager pass over the same workload: HIP events around every conv launch
        m2 = AudioModel(hp, device=dev, use_graph=False)
        m2._skip_exchange = True              # rank 0 only: the other ranks are already at the final barrier, a collective here would hang
        m2.set_inputs(s, mask, **inputs)
        for i in range(2):
            m2.optimize_parameters(i)
        torch.cuda.synchronize()
        kt = KernelTimer(lib)
        kt.install()
        nprof = min(args.steps, 3 if av else 10)
        for i in range(nprof):
            m2.optimize_parameters(i)
        fam = kt.summary()
        if args.layers:
            print("\n".join(kt.per_layer()), file=sys.stderr)
        kt.uninstall()
        tot_t = sum(v[1] for v in fam.values())
        tot_f = sum(v[0] for v in fam.values())
        # dominant kernel = the MFMA kernel family that holds the most time of the step, whichever it is; each family is priced
        # against the ceiling of ITS arithmetic: f16x2 = 2500 / 3 partial products, bf16x3 = 2500 / 6, exact fp32 MFMA = 157.3
        cands = [k for k in fam if k != "direct"]                      # "direct" = the Cin = 1 / Cout = 1 streaming convs (HBM-bound, no MFMA)
        dom = max(cands, key=lambda k: fam[k][1])
        f, t, n = fam[dom]
        ach = f / t * 1e-12
        peak = peak_of(dom)
        nprod = 3 if dom.endswith("_f16x2") else 6
        # the same kernel with nothing running beside it (weight gradients back on the main stream): what the kernel
        # itself reaches, without the time-sharing the as-run figure above includes
        alone = step_floor = None
        if m2._wgrad_stream is not None and not av:
            side, m2._wgrad_stream = m2._wgrad_stream, None
            dreal, m2._dreal_stream = m2._dreal_stream, None
            m2.optimize_parameters(0)
            torch.cuda.synchronize()
            kt1 = KernelTimer(lib)
            kt1.install()
            for i in range(min(nprof, 5)):
                m2.optimize_parameters(i)
            fam1 = kt1.summary()
            f1, t1, n1 = fam1[dom]
            kt1.uninstall()
            sf = StepFloor(lib)
            sf.install()
            for i in range(3):
                m2.optimize_parameters(i)
            step_floor = sf.summary(3, ms_per_step)
            sf.uninstall()
            m2._wgrad_stream, m2._dreal_stream = side, dreal
            alone = {"achieved": round(f1 / t1 * 1e-12, 2), "frac": round(f1 / t1 * 1e-12 / peak, 4), "avg_launch_us": round(t1 / n1 * 1e6, 2),
                     "note": "single-stream pass (weight gradients and D(real) back on the main stream): every launch with the chip to itself -- what the "
                             "kernels reach, without the time-sharing the as-run figures include (the side streams' kernels run beside the main "
                             "chain by design, on sub-CU grids)",
                     "frac_by_kernel": {k: round(v[0] / v[1] * 1e-12 / peak_of(k), 3) for k, v in sorted(fam1.items()) if v[1] > 0 and k != "direct"},
                     "conv_ms_per_step": round(sum(v[1] for v in fam1.values()) / min(nprof, 5) * 1e3, 3)}
        out["roofline"] = {
            "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": None,
            "peak_note": ("algorithmic fp32 flops against the dense 16-bit MFMA peak (2500) / %d partial products per MAC; "
                          "the same flops are %.2fx the fp32-MFMA peak (157.3); with random operands the fp16 pipes sustain 1677 at the clock the power budget "
                          "leaves (profiles/r04_d_mfma_shapes.txt; bf16: 1810, profiles/r01_e_mfma_probe.txt), i.e. %.2f of the sustained ceiling" % (nprod, ach / MFMA_F32_PEAK_TFLOPS, ach / (1677.0 / nprod)))
                         if peak != MFMA_F32_PEAK_TFLOPS else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)",
            "dominant_rule": "the MFMA kernel family with the largest share of the summed conv-launch time of the step (conv_time_share_by_kernel); "
                             "frac_by_kernel prices every family against its own ceiling; families are the ones the library reports per call "
                             "(viai_conv2d_last_kernel: the name's suffix is the arithmetic), descriptions in `kernels`",
            "kernel": FAMILY_KERNELS.get(dom, dom), "family": dom, "launches_per_step": n // nprof, "avg_launch_us": round(t / n * 1e6, 2),
            "algorithmic_gflop_per_step_in_kernel": round(f / nprof * 1e-9, 1), "algorithmic_gflop_per_launch": round(f / n * 1e-9, 2),
            "how": "HIP events on the launch stream of each call (main stream, or the weight-gradient side stream), %d instrumented steps of the same eager step after the timed region" % nprof,
            "kernels": {k: FAMILY_KERNELS.get(k, k) for k in sorted(fam)},
            "conv_time_share_by_kernel": {k: round(v[1] / tot_t, 3) for k, v in sorted(fam.items())},
            "conv_tflops_by_kernel": {k: round(v[0] / v[1] * 1e-12, 2) for k, v in sorted(fam.items()) if v[1] > 0},
            "frac_by_kernel": {k: round(v[0] / v[1] * 1e-12 / peak_of(k), 3) for k, v in sorted(fam.items()) if v[1] > 0 and k != "direct"},
            "conv_frac_whole_step": round(tot_f / tot_t * 1e-12 / (MFMA_BF16_PEAK_TFLOPS / 3.0), 4),
            "conv_ms_per_step": round(tot_t / nprof * 1e3, 3),
            "conv_gflop_per_step_timed": round(tot_f / nprof * 1e-9, 1),
        }
        if "step_tflops" not in out["config"]:                                        # other shapes / configs: what the instrumented pass summed
            out["config"]["algorithmic_gflop_per_step"] = round(tot_f / nprof * 1e-9, 1)
            out["config"]["step_tflops"] = round(tot_f / nprof * 1e-12 / (ms_per_step * 1e-3), 2)
            out["config"]["algorithmic_gflop_note"] = "2 x MACs of every conv launch of the step (forward, data gradient, weight gradient), summed by the instrumented pass"
        if alone is not None:
            out["roofline"]["standalone"] = alone
        if step_floor is not None:
            out["roofline"]["step_floor"] = step_floor
            out["roofline"]["step_floor_ms"] = step_floor["step_floor_ms"]
            out["roofline"]["frac_of_floor"] = step_floor["frac_of_floor"]
            out["roofline"]["step_floor_at_peak_ms"] = step_floor["step_floor_at_peak_ms"]
            out["roofline"]["frac_of_floor_at_peak"] = step_floor["frac_of_floor_at_peak"]
        if world == 1 and not av and BF3:
            out["roofline"]["power_limit"] = dvfs_probe(dev)
        # memory-side traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process, so the
        # number comes from the committed counter summary of the same kernel on its largest layer (D.conv3)
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_dconv3.json")))
        pmc = pmcs[-1] if pmcs else ""
        if BF3 and pmc and not av:
            want = PMC_KERNEL_OF.get(dom)
            if want:
                for k, v in json.load(open(pmc)).items():
                    if want in k and "hbm_bytes" in v:
                        out["roofline"]["traffic"] = round(v["hbm_bytes"])
                        out["roofline"]["traffic_source"] = "committed: profiles/%s (rocprofv3 --pmc, 2*FETCH_SIZE+WRITE_SIZE per launch, D.conv3)" % os.path.basename(pmc)
                        out["roofline"]["traffic_note"] = (
                            "bytes per launch of this kernel on D.conv3 (algorithmic: x 33.6 MB + dy 67.1 MB + dw 4.7 MB for the weight gradient; 105.4 MB = in + out + weights for "
                            "forward / data gradient): 2*FETCH_SIZE + WRITE_SIZE from profiles/%s (tools/profile_layer.py under rocprofv3 --pmc); these L2 memory-side "
                            "counters include Infinity-Cache hits, i.e. they are L2-miss traffic, an upper bound on HBM bytes") % os.path.basename(pmc)
                        break
        # the vision-infused step: the dominant family's kernel on the ResNet stage it spends most time in (layer1 for the 64-channel instances, layer2 otherwise)
        pmcs_av = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_resnet_layer2.json")))
        if BF3 and av and pmcs_av:
            want = PMC_KERNEL_OF.get(dom)
            if want:
                for k, v in json.load(open(pmcs_av[-1])).items():
                    if want in k and "hbm_bytes" in v:
                        out["roofline"]["traffic"] = round(v["hbm_bytes"])
                        out["roofline"]["traffic_source"] = "committed: profiles/%s (rocprofv3 --pmc, 2*FETCH_SIZE+WRITE_SIZE per launch, ResNet layer2)" % os.path.basename(pmcs_av[-1])
                        out["roofline"]["traffic_note"] = (
                            "bytes per launch of this kernel on ResNet layer2's 3x3 conv (1024 frames x 28 x 28 x 128 -> 128; algorithmic: in 411 MB + out 411 MB + weights 0.6 MB "
                            "for forward / data gradient, x 411 MB + dy 411 MB + dw 0.6 MB x slabs for the weight gradient): 2*FETCH_SIZE + WRITE_SIZE from profiles/%s "
                            "(tools/profile_layer.py under rocprofv3 --pmc); L2-miss traffic, an upper bound on HBM bytes") % os.path.basename(pmcs_av[-1])
                        break
        steps_pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_step.json")))
        if BF3 and steps_pmc and not av:
            ps = json.load(open(steps_pmc[-1]))
            out["roofline"]["mfma_busy"] = {"conv_kernels_time_weighted": ps["mfma_busy_conv_kernels_time_weighted"], "whole_step": ps["mfma_busy_whole_step"],
                                            "dominant_kernel": next((v["mfma_busy"] for k, v in ps["kernels"].items() if PMC_KERNEL_OF.get(dom, "?").replace(" ", "") in k.replace(" ", "")), None),
                                            "source": "committed: profiles/%s (tools/pmc_step.sh, not measured in this run)" % os.path.basename(steps_pmc[-1])}
        del m2
    if rank == 0 and world == 1 and not av and not args.no_roofline:            # (N = 1 only: at N > 1 the other ranks wait in the final barrier)
        out["stages"] = front_end_stages(dev, args.batch, args.bins, args.frames)
    if rank == 0 and world == 1 and not av and not args.no_roofline:
        out["config"]["launch_modes"] = launch_modes(hp, dev, s, mask)
        out["config"]["host_enqueue_note"] = ("host_enqueue_ms_per_step is the LOOP figure: the host runs ahead until the device queues fill and then "
                                              "waits inside a launch, so it tracks the device time in every launch mode; the host's own cost per step is "
                                              "launch_modes.*.host_ms_per_step_idle_queue (one step enqueued into an idle queue): eager vs the C launch "
                                              "plan (bench.py --plan; bitwise the eager step)")
    if rank == 0 and world == 1 and not av and not args.no_roofline and not args.no_extra and (args.batch, args.bins, args.frames) == (16, 256, 256):
        del model
        model = None
        torch.cuda.empty_cache()
        out["extra"] = extra_legs()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_av(args, hp.num_D) if av else cpu_baseline(args)
    if rank == 0:
        emit(out, args)
    del model
    shutdown(world)


if __name__ == "__main__":
    main()
