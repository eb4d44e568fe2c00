// Standalone timing / phase-stamp harness for the LDS-DMA halo kernels (csrc/conv_halo_dma.hip): no torch, no library -- the kernel file is
// compiled into this probe.  Results are NOT checked here (tests/test_p16_gpu.py holds the kernel to the register-staged one bit for bit);
// operands are random fp16 bit patterns with sane exponents.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DVIAI_PROF] -o halo_dma_bench tools/probes/halo_dma_bench.hip
//   ./halo_dma_bench N H W [iters]
#include "../../vision-infused-audio-inpainter-viai_amd/csrc/conv_halo_dma.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
int viai_plan_log_on = 0;
void viai_plan_note(const void*, void*, dim3, dim3, const unsigned char*, const unsigned*, int) {}
thread_local ViaiKernelTag viai_kernel_tag = {nullptr, 0};

static void fill_f16(std::vector<unsigned short>& v, unsigned seed) {
    for (auto& x : v) { seed = seed * 1664525u + 1013904223u; x = (unsigned short)(((seed >> 9) & 0x83ffu) | 0x3000u | ((seed >> 3) & 0x0c00u)); }   // |x| in [2^-3, 2^0)
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16, H = argc > 2 ? atoi(argv[2]) : 128, W = argc > 3 ? atoi(argv[3]) : 128, iters = argc > 4 ? atoi(argv[4]) : 20;
    const size_t px = (size_t)N * H * W;
    std::vector<unsigned short> hx(px * 64), hw(9 * 2 * 2 * 64 * 8);
    fill_f16(hx, 1); fill_f16(hw, 2);
    unsigned short *dx, *dw; float *dy, *dstat, *damax;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dy, px * 32 * 4); hipMalloc(&dstat, 2 * 32 * (px / 128) * 4); hipMalloc(&damax, 4);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    const float am = 1.0f; hipMemcpy(damax, &am, 4, hipMemcpyHostToDevice);
    ConvArgs a{};
    a.in = (const float*)dx; a.wp = (const float*)dw; a.out = dy; a.stat = dstat; a.C1 = 32; a.Cout = 32; a.OC1 = 32; a.M = (int)px; a.amax = damax; a.in_p16 = 1;
    a.g.N = N; a.g.IH = a.g.OH = a.g.SH = H; a.g.IW = a.g.OW = a.g.SW = W; a.g.ly = a.g.lx = a.g.my = a.g.mx = 1; a.g.ntaps = a.g.wtaps = 9;
    int slots[9]; for (int t = 0; t < 9; ++t) slots[t] = t;
    setenv("VIAI_HALO_DMA", "1", 1);
#ifdef VIAI_PROF
    unsigned long long* dprof; hipMalloc(&dprof, 512 * 16 * 8 * 8); hipMemset(dprof, 0, 512 * 16 * 8 * 8);
    viai_dma_prof_buf = dprof;
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) if (int e = viai_conv_halo_c32_dma_launch(a, -1, -1, slots, 0)) { printf("launch error %d\n", e); return 1; }
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        hipEventRecord(e0, 0); viai_conv_halo_c32_dma_launch(a, -1, -1, slots, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)px * 256;
    printf("c32 dma  %d x %d x %d : median %.1f us  min %.1f us  (%.2f TB/s at the median; %d launches; last error %d)\n", N, H, W, ts[ts.size() / 2], ts[0], bytes / ts[ts.size() / 2] * 1e-6, iters, (int)hipGetLastError());
#ifdef VIAI_PROF
    std::vector<unsigned long long> hp(512 * 16 * 8); hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost);
    const char* names[7] = {"wait at barrier", "issue DMA", "MFMA phase", "vmcnt wait", "(decode)", "stores", "stats"};
    const int nblk = std::min(512, (int)(px / 128));
    for (int k = 0; k < 6; ++k) {
        double acc[7] = {0}; int cnt = 0;
        for (int b = 0; b < nblk; ++b) {
            const unsigned long long* s = &hp[((size_t)b * 16 + k) * 8];
            if (s[6] == 0) continue;
            for (int i = 0; i < 6; ++i) acc[i] += (double)(s[i + 1] - s[i]);
            ++cnt;
        }
        if (!cnt) break;
        printf("tile %d of a block (%d blocks), shader-clock cycles:", k, cnt);
        for (int i = 0; i < 6; ++i) printf("  %s %.0f", names[i], acc[i] / cnt);
        printf("\n");
    }
    { double a0 = 0, span = 0; int cnt = 0; for (int b = 0; b < nblk; ++b) { const unsigned long long* s = &hp[(size_t)b * 16 * 8]; if (s[0] && s[7]) { a0 += (double)(s[0] - s[7]); ++cnt; } 
        unsigned long long last = 0; for (int k = 0; k < 16; ++k) if (hp[((size_t)b * 16 + k) * 8 + 6]) last = hp[((size_t)b * 16 + k) * 8 + 6]; if (last) span += (double)(last - s[7]); }
      printf("prologue wait (first data + filter) -> first tile: %.0f cycles; first barrier -> last stamp: %.0f cycles\n", a0 / cnt, span / cnt); }
#endif
    return 0;
}
