// Standalone timing / stage-stamp harness for wgrad_patch_f16_kernel (csrc/conv_wgrad_patch.hip) on P16 operands.  Results are not checked here (tests/test_p16_gpu.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DVIAI_PROF] -o wgrad_patch_bench tools/probes/wgrad_patch_bench.hip ;  ./wgrad_patch_bench N H W Cin Cout [iters]
#include "../../vision-infused-audio-inpainter-viai_amd/csrc/conv_wgrad_patch.hip"
#include <cstdio>
#include <vector>
int viai_plan_log_on = 0;
void viai_plan_note(const void*, void*, dim3, dim3, const unsigned char*, const unsigned*, int) {}
thread_local ViaiKernelTag viai_kernel_tag = {nullptr, 0};
static void fill_f16(std::vector<unsigned short>& v, unsigned seed) {
    for (auto& x : v) { seed = seed * 1664525u + 1013904223u; x = (unsigned short)(((seed >> 9) & 0x83ffu) | 0x3000u | ((seed >> 3) & 0x0c00u)); }
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1024, H = argc > 2 ? atoi(argv[2]) : 28, W = argc > 3 ? atoi(argv[3]) : 28, Ci = argc > 4 ? atoi(argv[4]) : 128, Co = argc > 5 ? atoi(argv[5]) : 128;
    const int iters = argc > 6 ? atoi(argv[6]) : 10;
    const size_t px = (size_t)N * H * W;
    std::vector<unsigned short> hx(px * Ci * 2), hd(px * Co * 2);
    fill_f16(hx, 1); fill_f16(hd, 2);
    unsigned short *dx, *dd; float *ws, *damax;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dd, hd.size() * 2); hipMalloc(&damax, 8);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dd, hd.data(), hd.size() * 2, hipMemcpyHostToDevice);
    const float am[2] = {1.0f, 1.0f}; hipMemcpy(damax, am, 8, hipMemcpyHostToDevice);
    WgradArgs a{};
    a.x = (const float*)dx; a.dy = (const float*)dd; a.C1 = Ci; a.Cout = Co; a.M = (int)px; a.amax = damax; a.xmax = damax + 1; a.dy_p16 = 1; a.x_p16 = 1;
    a.g.N = N; a.g.IH = H; a.g.IW = W; a.g.OH = a.g.SH = H; a.g.OW = a.g.SW = W; a.g.ly = a.g.lx = 1; a.g.my = a.g.mx = 1; a.g.ntaps = a.g.wtaps = 9;
    for (int t = 0; t < 9; ++t) { a.g.dy[t] = t / 3 - 1; a.g.dx[t] = t % 3 - 1; a.g.ws[t] = t; }
    const int ks = viai_wgrad_patch_ksplit(a.g, Co, Ci, 0);
    hipMalloc(&ws, (size_t)ks * 9 * Co * Ci * 4);
    a.ws = ws;
#ifdef VIAI_PROF
    const size_t pn = 1024 * 32 * 4;
    unsigned long long* dprof; hipMalloc(&dprof, pn * 8); hipMemset(dprof, 0, pn * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(wg_prof_buf), &dprof, sizeof(dprof));
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) if (int e = viai_wgrad_patch_launch(a, 0)) { printf("launch error %d\n", e); return 1; }
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) viai_wgrad_patch_launch(a, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, gf = 2.0 * px * Co * 9 * Ci * 1e-9;
    printf("wgrad patch  %d x %d x %d : %d -> %d  slabs %d : %.1f us per launch  %.0f TFLOP/s  last error %d\n", N, H, W, Ci, Co, ks, us, gf / us * 1e3, (int)hipGetLastError());
#ifdef VIAI_PROF
    std::vector<unsigned long long> hp(pn); hipMemcpy(hp.data(), dprof, pn * 8, hipMemcpyDeviceToHost);
    for (int q = 0; q < 32; ++q) {
        double acc[2] = {0, 0}; int cnt = 0;
        for (int b = 0; b < 1024; ++b) { const unsigned long long* s = &hp[((size_t)b * 32 + q) * 4]; if (!s[2]) continue; acc[0] += (double)(s[1] - s[0]); acc[1] += (double)(s[2] - s[1]); ++cnt; }
        if (!cnt) break;
        printf("stage %2d (%4d blocks), ticks of s_memtime:  k-steps + staging %.0f   barrier wait %.0f\n", q, cnt, acc[0] / cnt, acc[1] / cnt);
    }
#endif
    return 0;
}
