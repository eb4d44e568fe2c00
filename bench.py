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
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

METRIC = "masked mel-spectrogram clips/sec (G+D train step, 256x256 b16) at 1/2/4/8 GPUs"
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
# The default conv math is the bf16x3 split (csrc/conv_igemm_bf3.hip): every fp32 MAC costs six bf16 MFMA MACs, so the
# ceiling for ALGORITHMIC fp32 flops on that kernel is the bf16 peak / 6.  VIAI_MATH=fp32 selects the exact-fp32 MFMA kernel.
BF3 = os.environ.get("VIAI_MATH", "") != "fp32"
F16X2 = BF3 and os.environ.get("VIAI_F16X2", "1") != "0"     # wide forward layers: f16x2 split (three partial products)
PEAK = MFMA_BF16_PEAK_TFLOPS / 6.0 if BF3 else MFMA_F32_PEAK_TFLOPS
DOMINANT = ("conv_igemm_bf3_frag_kernel<3,2,2,2,2> (128x128x32 bf16x3 split-MFMA implicit-GEMM conv, fp32-grade accuracy; data-gradient launches"
            + ("" if F16X2 else " and forward launches") + ")"
            if BF3 else "conv_igemm_kernel<32,2,2,2,2> (128x128x32 fp32-MFMA implicit-GEMM conv, fwd + dgrad)")


# kernel behind each launch family of KernelTimer (the mirror predicates above decide the family of a call)
FAMILY_KERNELS = {
    "wgrad_patch_f16x2": "wgrad_patch_f16_kernel<1,128,64,4,1> (csrc/conv_wgrad_patch.hip: weight gradient of the stride-1 3x3 layers with >= 128 x 64 channels; all nine taps per block, "
                         "dy rows + x patch staged once per 64 pixels as [pixel][channel] fp16 planes, MFMA operands through ds_read_b64_tr_b16; f16x2 split)",
    "wgrad_patch_narrow_f16x2": "wgrad_patch_f16_kernel<1,32,32,4,4> (the narrow instance: 32 x 32 channel tile, four waves split the tile rows of a stage and write one split-K slab each)",
    "wgrad_patch_s2_f16x2": "wgrad_patch_f16_kernel<2,128,32,2,1> (the stride-2 instance: 128 x 32 channel tile, input patch as four parity sub-patches, 2 blocks / CU)",
    "wgrad_bf3_f16x2": "wgrad_bf3_kernel<2,TM,TN> (csrc/conv_wgrad_bf3.hip: weight gradient, one block per tap x Cout tile x Cin tile, tiles transposed into LDS; f16x2 split)",
    "wgrad_bf3": "wgrad_bf3_kernel<3,TM,TN> (bf16x3 weight gradient)",
    "wgrad_mfma": "wgrad_mfma_kernel<*> (exact fp32 MFMA weight gradient, <= 32-channel layers)",
    "wgrad32_all_taps": "wgrad32_halo_kernel (exact fp32 MFMA, <= 32 x <= 32 channels, stride 1: all taps per block)",
    "halo_wide256_f16x2": "conv_halo_wide_f16_kernel<2,4,2,2> (stride-1 3x3 conv, 8x16-pixel x 256-channel tile, f16x2 split MFMA: the 10x18 input patch of a 32-channel chunk is "
                          "staged once in LDS and read by all nine taps; weight fragments straight from global; forward and data-gradient launches of D.conv3)",
    "halo_wide128_f16x2": "conv_halo_wide_f16_kernel<2,2,2,2> (as above, 128-channel tile)",
    "halo_wide64_f16x2": "conv_halo_wide_f16_kernel<2,2,2,1> (64-channel tile)", "halo_wide32_f16x2": "conv_halo_wide_f16_kernel<4,1,1,1> (32-channel tile)",
    "halo_wide_s2_f16x2": "conv_halo_wide_f16_kernel<2,4,2,{2,1},2> (stride-2 forward, four parity sub-patches)",
    "halo_f16x2": "conv_halo_f16_c32_kernel / conv_halo_bf3_kernel<CIN,TN,2> (32/64-channel stride-1 layers, f16x2)",
    "halo": "conv_halo_bf3_kernel<CIN,TN,3> (32/64-channel stride-1 layers, bf16x3)",
    "dgrad_s2_f16x2": "conv_dgrad_s2_patch_kernel (3x3 stride-2 data gradient, four parity classes fused, dy patch staged once per 32-channel chunk; f16x2)",
    "dgrad_s2": "conv_dgrad_s2_bf3_kernel<3> (3x3 stride-2 data gradient, bf16x3)",
    "igemm128x256_f16x2": "conv_igemm_bf3_frag_kernel<2,2,2,2,4> (128x256x32 f16x2 gather-GEMM conv, eight waves)",
    "igemm128x128_f16x2": "conv_igemm_bf3_frag_kernel<2,2,2,2,2> (128x128x32 f16x2 gather-GEMM conv)",
    "igemm128x128": "conv_igemm_bf3_frag_kernel<3,2,2,2,2> (128x128x32 bf16x3 gather-GEMM conv)",
}
PMC_KERNEL_OF = {"wgrad_patch_f16x2": "wgrad_patch_f16_kernel<1, 128", "wgrad_patch_s2_f16x2": "wgrad_patch_f16_kernel<2", "wgrad_patch_narrow_f16x2": "wgrad_patch_f16_kernel<1, 32", "wgrad_bf3_f16x2": "wgrad_bf3_kernel", "halo_wide256_f16x2": "conv_halo_wide_f16_kernel",
                 "halo_wide128_f16x2": "conv_halo_wide_f16_kernel", "igemm128x256_f16x2": "frag_kernel<2, 2, 2, 2, 4", "igemm128x128_f16x2": "frag_kernel<2", "igemm128x128": "frag_kernel<3"}


def math_string():
    """what the conv kernels compute in, from the switches in effect (csrc/conv_api.hip reads the same environment)"""
    if not BF3:
        return "VIAI_MATH=fp32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32) in every conv kernel"
    f16b = os.environ.get("VIAI_F16_BACKWARD", "1") != "0"
    if F16X2:
        return ("fp32 tensors and fp32 accumulation; conv products on the fp16 matrix cores through the f16x2 operand split (two fp16 terms per fp32 "
                "operand = 22 significand bits, three partial products, power-of-two pre-scaling: static x16 for activations / x256 for weights, "
                "per-tensor from max|dy| for gradients) in the forward%s kernels of every layer with > 1 channel on both sides; bf16x3 (three bf16 "
                "terms, six partial products) where no BatchNorm produces the gradient scale%s; exact fp32 MFMA in the <= 32-channel weight gradients the patch kernel does not tile; "
                "plain fp32 FMA in the Cin = 1 / Cout = 1 streaming convs; measured 2.7-2.9e-7 relative vs fp64 per layer (CPU fp32: 1.8e-7); "
                "activations saturate at |x| > 4094 (tests/test_kernels_gpu.py::test_f16x2_saturates_instead_of_nan)"
                % (", data-gradient and weight-gradient" if f16b else "", "" if f16b else " and in every backward kernel (VIAI_F16_BACKWARD=0)"))
    return "fp32 tensors and fp32 accumulation; conv products on the bf16 matrix cores through the bf16x3 split (six partial products) (VIAI_F16X2=0)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
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
    ap.add_argument("--layers", action="store_true", help="print the per-layer conv timing table to stderr")
    ap.add_argument("--config", choices=["audio", "av", "av_msd"], default="audio",
                    help="audio = BASELINE configs[1] (the metric); av = configs[2] (+ResNet-18 visual branch, N = T/4 frames); "
                         "av_msd = configs[3] model (+3-scale D).  Non-default configs print the same JSON shape without roofline")
    ap.add_argument("--cpu-batch", type=int, default=2, help="clips per CPU-baseline step (bounded sample)")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of the conv launches on torch's current stream (the stream the kernels run on).
    Wraps the ctypes entry points of libviai_hip.so; flops are the ALGORITHMIC 2*MACs of each call."""

    def __init__(self, lib):
        self.lib = lib
        self.records = []          # (family, flops, n_launches, ev0, ev1)
        self.orig = {}

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

        def wrap(name, family_of, counts=lambda args: True):
            fn = getattr(lib, name)
            self.orig[name] = fn

            def timed(desc_ref, *args):
                d = desc_ref._obj
                fam, nl = family_of(d)
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
                self.records.append((fam, flops, nl, e0, e1, name, (d.N, d.IH, d.IW, d.C1 + d.C2, d.Cout, d.kh, d.kw, d.sh, d.sw, d.transposed)))
                return r
            setattr(lib, name, timed)

        def tile_m(M, n_out):                     # mirror of viai_igemm_tile_m (csrc/conv_igemm.hip)
            if n_out <= 32:
                return 128
            b = -(-M // 128) * (-(-n_out // 128) if n_out > 64 else -(-n_out // 64))
            return 128 if b >= 512 else 64

        def igemm_name(M, n_out):
            if BF3 and n_out > 32 and not (tile_m(M, n_out) == 128 and n_out > 64) and -(-M // 64) * -(-n_out // 64) < 512:
                return "igemm_sk32x32"               # mirror of viai_bf3_sk_ok (csrc/conv_igemm_bf3.hip)
            if tile_m(M, n_out) == 64:
                return "igemm64x64"
            return "igemm128x128" if n_out > 64 else ("igemm128x64" if n_out > 32 else "igemm128x32")

        def wide_f16(M, n_out):                   # mirror of viai_conv_igemm_bf3_launch: the eight-wave 128 x 256 tile where it fits
            wn4 = os.environ.get("VIAI_F16_WN4", "1") != "0"
            return "igemm128x256_f16x2" if (wn4 and n_out % 256 == 0 and -(-M // 128) * (n_out // 256) >= 256) else "igemm128x128_f16x2"

        def out_pixels(d):
            oh = (d.IH - 1 - 2 * d.ph + d.kh) if d.transposed else (d.IH + 2 * d.ph - d.kh) // d.sh + 1
            ow = (d.IW - 1 - 2 * d.pw + d.kw) if d.transposed else (d.IW + 2 * d.pw - d.kw) // d.sw + 1
            return d.N * oh * ow

        def halo_ok(c_in, c2, c_out, d, oh, ow):     # mirror of viai_conv_halo_ok (csrc/conv_halo_bf3.hip)
            return (BF3 and c2 == 0 and c_in in (32, 64) and c_out <= 64 and d.sh == 1 and d.sw == 1
                    and d.kh <= 3 and d.kw <= 3 and oh % 8 == 0 and ow % 16 == 0)

        def halo16(c_in, c_out, d):                  # mirror of viai_conv_halo16_ok: filter-in-registers f16x2 variant
            return F16X2 and os.environ.get("VIAI_HALO16", "1") != "0" and c_in == 32 and c_out <= 32 and d.kh == 3 and d.kw == 3

        def halo_wide(c1, c2, c_out, oc1_split_ok, d, oh, ow):        # mirror of viai_conv_halo_wide_ok (csrc/conv_halo_bf3.hip)
            if not F16X2 or os.environ.get("VIAI_HALO_WIDE", "1") == "0":
                return None
            if c1 % 32 or c2 % 32 or c1 < 32 or not oc1_split_ok or not (c_out in (32, 64) or c_out % 128 == 0):
                return None
            if c_out <= 64 and c1 + c2 <= 64 and c2 == 0:
                return None
            s2 = (d.sh, d.sw) == (2, 2) and not d.transposed and os.environ.get("VIAI_HALO_WIDE_S2", "1") != "0"
            if (d.kh, d.kw) != (3, 3) or not ((d.sh, d.sw) == (1, 1) or s2) or oh % 8 or ow % 16:
                return None
            if s2 and (c2 != 0 or c_out % 128):
                return None
            tiles = d.N * (oh // 8) * (ow // 16)
            if tiles * (c_out // 128 if c_out >= 128 else 1) < 192:
                if not (c_out >= 128 and tiles * (c_out // 64) >= int(os.environ.get("VIAI_HALO_WIDE_MIN64", "96"))):
                    return None
                return "halo_wide_s2_f16x2" if s2 else "halo_wide64_f16x2"
            if s2:
                return "halo_wide_s2_f16x2"
            if c_out <= 64:
                return "halo_wide%d_f16x2" % c_out
            wn4 = os.environ.get("VIAI_HALO_WIDE_WN4", "1") != "0" and c_out % 256 == 0 and tiles * (c_out // 256) >= 256
            return "halo_wide256_f16x2" if wn4 else "halo_wide128_f16x2"

        def out_hw(d):
            oh = (d.IH - 1 - 2 * d.ph + d.kh) if d.transposed else (d.IH + 2 * d.ph - d.kh) // d.sh + 1
            ow = (d.IW - 1 - 2 * d.pw + d.kw) if d.transposed else (d.IW + 2 * d.pw - d.kw) // d.sw + 1
            return oh, ow

        def fam_fwd(d):
            cin = d.C1 + d.C2
            if cin == 1 or d.Cout == 1:
                return "direct", 1
            if halo_ok(d.C1, d.C2, d.Cout, d, *out_hw(d)):
                return ("halo_f16x2" if F16X2 else "halo"), 1
            hw = halo_wide(d.C1, d.C2, d.Cout, True, d, *out_hw(d))
            if hw:
                return hw, 1
            nm = igemm_name(out_pixels(d), d.Cout)
            if F16X2 and nm == "igemm128x128":
                return wide_f16(out_pixels(d), d.Cout), 1                                               # conv_igemm_bf3_frag_kernel<2,...>
            return (nm + "_f16x2" if (F16X2 and os.environ.get("VIAI_F16_PLANAR", "1") != "0") else nm), 1    # planar f16x2 LDS-weight / split-K kernels

        def fam_dgrad(d):
            cin = d.C1 + d.C2
            if cin == 1 or d.Cout == 1:
                return "direct", 1
            if halo_ok(d.Cout, 0, cin, d, d.IH, d.IW):
                return "halo", 1
            if (BF3 and not d.transposed and (d.kh, d.kw, d.sh, d.sw, d.ph, d.pw) == (3, 3, 2, 2, 1, 1) and d.IH % 2 == 0 and d.IW % 2 == 0
                    and d.Cout % 32 == 0 and cin % 64 == 0 and -(-(d.N * (d.IH // 2) * (d.IW // 2)) // 128) * (cin // 64) >= (32 if ((d.IH // 2) % 8 == 0 and (d.IW // 2) % 16 == 0) else 256)):
                return "dgrad_s2", 1                 # mirror of viai_dgrad_s2_ok (csrc/conv_dgrad_s2_bf3.hip); f16x2: the patch-staged kernel where the base lattice tiles in 8 x 16
            ncls = d.sh * d.sw                        # one launch per output parity class
            return igemm_name(-(-(d.N * d.IH * d.IW) // ncls), cin), ncls

        def fam_wgrad(d):
            cin = d.C1 + d.C2
            if cin == 1 or d.Cout == 1:
                return "direct", 1
            if BF3 and cin > 32 and d.Cout > 32 and not (d.C2 > 0 and d.C1 % 64):
                return "wgrad_bf3", 1                # mirror of viai_wgrad_bf3_ok (csrc/conv_wgrad_bf3.hip)
            if (os.environ.get("VIAI_WGRAD32", "1") != "0" and d.C2 == 0 and cin <= 32 and d.Cout <= 32 and d.sh == 1 and d.sw == 1
                    and d.kh <= 3 and d.kw <= 3 and out_hw(d)[1] % 32 == 0):
                return "wgrad32_all_taps", 1         # mirror of viai_wgrad32_ok (csrc/conv_wgrad.hip)
            return "wgrad_mfma", 1

        def fam_dgrad_f16(d):                        # viai_conv2d_dgrad_f16: the f16x2 instances of the same kernels
            f, n = fam_dgrad(d)
            if f != "halo" and d.sh == 1 and d.sw == 1:
                hw = halo_wide(d.Cout, 0, d.C1 + d.C2, d.C2 == 0 or d.C1 % 32 == 0, d, d.IH, d.IW)
                if hw:
                    return hw, 1
            if f == "igemm128x128":
                return wide_f16(-(-(d.N * d.IH * d.IW) // n), d.C1 + d.C2), n
            return f + "_f16x2", n

        wrap("viai_conv2d_fwd", fam_fwd)
        wrap("viai_conv2d_dgrad", fam_dgrad)
        def fam_wgrad_f16(d):
            f, n = fam_wgrad(d)
            oh, ow = out_hw(d)
            # mirror of pick() in csrc/conv_wgrad_patch.hip
            s1 = (d.sh, d.sw) == (1, 1)
            s2 = (d.sh, d.sw) == (2, 2) and not d.transposed and os.environ.get("VIAI_WGRAD_PATCH_S2", "1") != "0"
            if (os.environ.get("VIAI_WGRAD_PATCH", "1") != "0" and (d.kh, d.kw) == (3, 3) and (s1 or s2)
                    and d.N * -(-oh // 8) * -(-ow // 16) >= 64 and oh * ow * 3 >= -(-oh // 8) * -(-ow // 16) * 128 and f not in ("direct",)):
                def tiles_ok(bm, bn):
                    return d.Cout % bm == 0 and d.C1 % bn == 0 and d.C2 % bn == 0 and d.C1 >= bn
                if s2 and tiles_ok(128, 32):
                    return "wgrad_patch_s2_f16x2", n
                if s1 and tiles_ok(128, 64):
                    return "wgrad_patch_f16x2", n
                if s1 and tiles_ok(32, 32) and os.environ.get("VIAI_WGRAD_PATCH_NARROW", "1") != "0":
                    return "wgrad_patch_narrow_f16x2", n
            return f + "_f16x2", n

        wrap("viai_conv2d_dgrad_f16", fam_dgrad_f16)
        wrap("viai_conv2d_wgrad_f16", fam_wgrad_f16)
        wrap("viai_conv2d_wgrad", fam_wgrad)
        # fused Cin = 1 conv + BatchNorm layer: statistics pass (z = NULL, args[6]) + apply pass; the weight gradient from dz
        wrap("viai_conv2d_cin1_bn_fwd", lambda d: ("direct", 1), counts=lambda args: bool(args[6]))
        wrap("viai_conv2d_cin1_bn_wgrad", lambda d: ("direct", 1))

    def per_layer(self):
        torch.cuda.synchronize()
        agg = {}
        for f, flops, nl, e0, e1, name, key in self.records:
            t = agg.setdefault((name, f, key), [0.0, 0.0, 0])
            t[0] += flops; t[1] += e0.elapsed_time(e1) * 1e-3; t[2] += 1
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        return ["%-18s %-13s %-44s n=%3d avg %8.1f us  %6.1f TF/s" % (k[0][5:], k[1], str(k[2]), v[2], v[1] / v[2] * 1e6, v[0] / v[1] * 1e-12)
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
           "sample": "oracle/viai_oracle.train_step (torch CPU fp32, %d threads of the %d logical CPUs this process may use: more threads are slower, "
                     "see bench.py), %d clips of %dx%d per step (the GPU line's workload), best of %d step(s) (%.1f s of CPU work), %.2f s/step"
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
    return {"stft_mel_gbps": round(b_fe / t_fe * 1e-9, 1), "stft_mel_us": round(t_fe * 1e6, 1), "stft_mel_frac_of_hbm_peak": round(b_fe / t_fe / 8e12, 4),
            "mask_gbps": round(b_mk / t_mk * 1e-9, 1), "mask_us": round(t_mk * 1e6, 1), "mask_frac_of_hbm_peak": round(b_mk / t_mk / 8e12, 4),
            "note": "HIP events around 50 back-to-back launches, inputs resident in HBM; algorithmic bytes: STFT->mel = waveform in (%d samples/clip) + mel out "
                    "(%d x %d), mask = mel in + out; both stages are far smaller than the GPU's caches at this batch, so the figure is launch-/latency-bound, "
                    "not a sustained-bandwidth measurement" % (n_samples, bins, fr)}


def main():
    args = parse()
    from viai_amd import _lib, ddp, synth
    from viai_amd.model import AudioModel, StepConfig

    rank, local, world = ddp.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _lib.load()

    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths, hp.batch_size = args.bins, args.frames, args.batch
    torch.manual_seed(1234)                               # identical init on every rank (DDP semantics)
    if args.config != "audio":
        hp.use_video, hp.lambda_contrast = True, 0.1
        hp.num_D = 3 if args.config == "av_msd" else 1
        args.no_roofline = args.no_cpu_baseline = True
    model = AudioModel(hp, device=dev, use_graph=args.graph, use_plan=args.plan)
    ddp.broadcast_arena(model.arena_G.flat)
    ddp.broadcast_arena(model.arena_D.flat)

    s = synth.mel_batch(args.batch, args.bins, args.frames, "bench.s", rank).to(dev)
    mask = synth.time_mask(args.batch, args.frames, "bench.mask", rank).to(dev)
    if args.config != "audio":
        nf = args.frames // 4
        video = synth.uniform("bench.video.r%d" % rank, (args.batch, nf, 3, 224, 224), -1, 1).to(dev)
        flow = synth.uniform("bench.flow.r%d" % rank, (args.batch, nf, 2, 224, 224), -1, 1).to(dev)
        model.set_inputs(s, mask, video=video, flow=flow)
    else:
        model.set_inputs(s, mask)

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
        "config": {"workload": ("configs[1]: audio-only MelEncoder+MelDecoder G + MelDiscriminator (PatchGAN) D train step, "
                                "%dx%d mel, batch %d per GPU, BCE-GAN + 100*L1, Adam(2e-4, 0.5, 0.999)" % (args.bins, args.frames, args.batch))
                   if args.config == "audio" else
                   ("configs[%d] (NOT the metric config): vision-infused G (2x ResNet-18 on %d frames/clip, tiled into the bottleneck) + %s, "
                    "%dx%d mel, batch %d per GPU" % (2 if args.config == "av" else 3, args.frames // 4,
                                                      "PatchGAN D" if args.config == "av" else "3-scale D", args.bins, args.frames, args.batch)),
                   "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                   "launch": ("launch plan replayed from C (3 segments; same kernels / streams / edges as the eager step)" if args.plan else
                              "hipGraph replay (3 segments)" if args.graph else "eager, weight gradients on a side stream"),
                   "math": math_string(),
                   "host_enqueue_ms_per_step": round(enqueue_ms / args.steps, 3), "algorithmic_gflop_per_step": 1208.0, "step_tflops": round(1208.0 * 1e-3 / (ms_per_step * 1e-3), 2),
                   "loss_d": round(losses[0], 5), "loss_g": round(losses[1], 5)},
    }
    if comm_exposed is not None:
        out["config"]["comm_ms_exposed"] = comm_exposed
        out["config"]["exchange"] = ("RCCL all-reduce of the flat gradient arenas in buckets launched from gradient-ready hooks inside the backward "
                                     "(D: conv3..conv4 early, rest at the end; G: two decoder buckets early, encoder at the end); the G exchange + "
                                     "Adam(E,G) + weight re-pack overlap the next step's D(real) forward/backward; comm_ms_exposed = ms_per_step - "
                                     "ms_per_step of the same steps without the collectives")

    if rank == 0 and not args.no_roofline:
        # instrumented eager pass over the same workload: HIP events around every conv launch
        m2 = AudioModel(hp, device=dev, use_graph=False)
        m2._skip_exchange = True              # rank 0 only: the other ranks are already at the final barrier, a collective here would hang
        m2.set_inputs(s, mask)
        for i in range(2):
            m2.optimize_parameters(i)
        torch.cuda.synchronize()
        kt = KernelTimer(lib)
        kt.install()
        nprof = min(args.steps, 10)
        for i in range(nprof):
            m2.optimize_parameters(i)
        fam = kt.summary()
        if args.layers:
            print("\n".join(kt.per_layer()), file=sys.stderr)
        kt.uninstall()
        tot_t = sum(v[1] for v in fam.values())
        # dominant kernel = the MFMA kernel family that holds the most time of the step, whichever it is; each family is priced
        # against the ceiling of ITS arithmetic: f16x2 = 2500 / 3 partial products, bf16x3 = 2500 / 6, exact fp32 MFMA = 157.3
        def peak_of(k):
            if k in ("wgrad_mfma", "wgrad32_all_taps") or not BF3:
                return MFMA_F32_PEAK_TFLOPS
            return MFMA_BF16_PEAK_TFLOPS / 3.0 if k.endswith("f16x2") else MFMA_BF16_PEAK_TFLOPS / 6.0
        cands = [k for k in fam if k != "direct"]                      # "direct" = the Cin = 1 / Cout = 1 streaming convs (HBM-bound, no MFMA)
        dom = max(cands, key=lambda k: fam[k][1])
        f, t, n = fam[dom]
        ach = f / t * 1e-12
        peak = peak_of(dom)
        dom_name = FAMILY_KERNELS.get(dom, dom)
        # the same kernel with nothing running beside it (weight gradients back on the main stream): what the kernel
        # itself reaches, without the time-sharing the as-run figure above includes
        alone = None
        if m2._wgrad_stream is not None:
            side, m2._wgrad_stream = m2._wgrad_stream, None
            m2.optimize_parameters(0)
            torch.cuda.synchronize()
            kt1 = KernelTimer(lib)
            kt1.install()
            for i in range(min(nprof, 5)):
                m2.optimize_parameters(i)
            f1, t1, n1 = kt1.summary()[dom]
            kt1.uninstall()
            m2._wgrad_stream = side
            alone = {"achieved": round(f1 / t1 * 1e-12, 2), "frac": round(f1 / t1 * 1e-12 / peak, 4), "avg_launch_us": round(t1 / n1 * 1e6, 2),
                     "note": "single-stream pass: the dominant kernel without the concurrent weight-gradient stream"}
        out["roofline"] = {
            "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": None,
            "peak_note": ("algorithmic fp32 flops against the dense 16-bit MFMA peak (2500) / %d partial products per MAC; "
                          "the same flops are %.2fx the fp32-MFMA peak (157.3); with random operands the pipes sustain 1810 "
                          "(profiles/r01_e_mfma_probe.txt), i.e. %.2f of the sustained ceiling"
                          % (3 if dom.endswith("f16x2") else 6, ach / MFMA_F32_PEAK_TFLOPS, ach / (1810.0 / (3.0 if dom.endswith("f16x2") else 6.0))))
                         if peak != MFMA_F32_PEAK_TFLOPS else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)",
            "dominant_rule": "the MFMA kernel family with the largest share of the summed conv-launch time of the step (conv_time_share_by_kernel); "
                             "frac_by_kernel prices every family against its own ceiling",
            "kernel": dom_name, "launches_per_step": n // nprof, "avg_launch_us": round(t / n * 1e6, 2),
            "algorithmic_gflop_per_step_in_kernel": round(f / nprof * 1e-9, 1),
            "how": "HIP events on the launch stream of each call (main stream, or the weight-gradient side stream), %d instrumented steps of the same eager step after the timed region" % nprof,
            "launch_family_rule": ("igemm128x128 = conv_igemm_bf3_frag_kernel<3,2,2,2,2> (bf16x3); igemm128x128_f16x2 / igemm128x256_f16x2 = conv_igemm_bf3_frag_kernel<2,2,2,2,2> / <2,2,2,2,4> (f16x2 split: ceiling 2500/3; the 128x256 eight-wave tile where Cout % 256 == 0 and it still yields >= 256 blocks); igemm64x64 = conv_igemm_bf3_lds_kernel<1,1,2,2[,NP]>; igemm_sk32x32 = conv_igemm_bf3_sk_kernel<NP> (small-M layers, waves split K); a _f16x2 suffix on these = the NP = 2 instance (planar fp16 weight planes); "
                                   "igemm128x64/x32 = conv_igemm_bf3_lds_kernel<2,1,2,2>/<1,1,4,1>; halo_wide{256,128,64,32}_f16x2 = conv_halo_wide_f16_kernel<2,4,2,2> / <2,2,2,2> / <2,2,2,1> / <4,1,1,1> (stride-1 3x3 layers with Cin >= 32: patch staged once per 32-channel chunk), halo_wide_s2_f16x2 = its <2,4,2,{2,1},2> instances (stride-2 forward, four parity sub-patches); halo[_f16x2] = conv_halo_bf3_kernel<CIN,TN,3|2> (32/64-channel stride-1 layers) and, for 32 -> <=32 channels, conv_halo_f16_c32_kernel (filter in registers); dgrad_s2[_f16x2] = conv_dgrad_s2_bf3_kernel<3|2> / conv_dgrad_s2_patch_kernel (3x3 stride-2 data gradient, four parity classes fused; the patch kernel stages the dy patch once per 32-channel chunk); wgrad_bf3 = wgrad_bf3_kernel<*>; "
                                   "wgrad_mfma = fp32 wgrad_mfma_kernel<*> (<= 32-channel layers); wgrad32_all_taps = wgrad32_halo_kernel (fp32 MFMA, <= 32 x <= 32 channels, stride 1: all taps per block)") if BF3 else
                                  "igemm64x64 = conv_igemm_kernel<32,1,1,2,2> (small-M layers), igemm128xN = <32,2,2,2,2>/<32,2,1,2,2>/<32,1,1,4,1>",
            "conv_time_share_by_kernel": {k: round(v[1] / tot_t, 3) for k, v in sorted(fam.items())},
            "conv_tflops_by_kernel": {k: round(v[0] / v[1] * 1e-12, 2) for k, v in sorted(fam.items()) if v[1] > 0},
            "frac_by_kernel": {k: round(v[0] / v[1] * 1e-12 / peak_of(k), 3) for k, v in sorted(fam.items()) if v[1] > 0 and k != "direct"},
            "conv_frac_whole_step": round(sum(v[0] for v in fam.values()) / tot_t * 1e-12 / (MFMA_BF16_PEAK_TFLOPS / 3.0), 4),
            "conv_ms_per_step": round(tot_t / nprof * 1e3, 3),
        }
        if alone is not None:
            out["roofline"]["standalone"] = alone
        # memory-side traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process, so the
        # number comes from the committed counter summary of the same kernel on its largest layer (D.conv3)
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_dconv3.json")))
        pmc = pmcs[-1] if pmcs else ""
        if BF3 and pmc:
            want = PMC_KERNEL_OF.get(dom)
            if want:
                for k, v in json.load(open(pmc)).items():
                    if want in k and "hbm_bytes" in v:
                        out["roofline"]["traffic"] = round(v["hbm_bytes"])
                        out["roofline"]["traffic_note"] = (
                            "bytes per launch of this kernel on D.conv3 (algorithmic: x 33.6 MB + dy 67.1 MB + dw 4.7 MB for the weight gradient; 57 MB in + 50 MB out for "
                            "forward / data gradient): 2*FETCH_SIZE + WRITE_SIZE from profiles/%s (tools/profile_layer.py under rocprofv3 --pmc); these L2 memory-side "
                            "counters include Infinity-Cache hits, i.e. they are L2-miss traffic, an upper bound on HBM bytes") % os.path.basename(pmc)
                        break
        del m2
    if rank == 0 and args.config == "audio" and not args.no_roofline:
        out["stages"] = front_end_stages(dev, args.batch, args.bins, args.frames)
    if rank == 0 and world == 1 and args.config == "audio" and not args.no_roofline:
        out["config"]["launch_modes"] = launch_modes(hp, dev, s, mask)
        out["config"]["host_enqueue_note"] = ("host_enqueue_ms_per_step is the LOOP figure: the host runs ahead until the device queues fill and then "
                                              "waits inside a launch, so it tracks the device time in every launch mode; the host's own cost per step is "
                                              "launch_modes.*.host_ms_per_step_idle_queue (one step enqueued into an idle queue): eager vs the C launch "
                                              "plan (bench.py --plan; bitwise the eager step)")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()
        # leave without tearing the communicator down: destroy_process_group() on an RCCL group aborts the process now and then on
        # this stack (seen in the test suite), and a non-zero exit of one rank after the line is printed would read as a failed run
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
