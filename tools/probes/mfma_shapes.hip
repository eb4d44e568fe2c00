// Sustained rate of the fp16 MFMA shapes under RANDOM operands (the chip clocks to its power budget: MI355X_MICROARCH.md "DVFS give-back"): is the power per
// flop of v_mfma_f32_16x16x32_f16 (a quarter of the accumulator registers per instruction, twice the operand reads per flop) lower than 32x32x16's?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_shapes.bin tools/probes/mfma_shapes.hip && tools/probes/mfma_shapes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline void fill(f16x8& a, f16x8& b, unsigned& h, int dense, float bscale) {
    for (int e = 0; e < 8; ++e) {
        h = h * 1664525u + 1013904223u;
        float ra = dense ? ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) : 0.f;
        h = h * 1664525u + 1013904223u;
        float rb = dense ? ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * bscale : 0.f;
        a[e] = (_Float16)ra; b[e] = (_Float16)rb;
    }
}
// MODE 0: 32x32x16, 4 accumulators (64 registers);  MODE 1: 16x16x32, 16 accumulators (64 registers): the same flops per loop trip
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, int dense) {
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) fill(a[i], b[i], h, dense, 0.01f);
    float s = 0.f;
    if constexpr (MODE == 0) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[(i * 3 + r) & 3], acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 3], b[((i >> 2) + r) & 3], acc[i], 0, 0, 0);
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int dense) {
    const int iters = 4000, blocks = 512;
    float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 10, dense);
    (void)hipDeviceSynchronize();
    float best = 1e30f, worst = 0.f;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        probe<MODE><<<blocks, 256>>>(out, iters, dense);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        if (ms > worst) worst = ms;
    }
    const double flops = (double)blocks * 4 * iters * (MODE == 0 ? 32 * 32768.0 : 64 * 16384.0);
    printf("%s, %s operands, 2 blocks/CU: best %.1f TFLOP/s, slowest of 8 %.1f TFLOP/s\n", name, dense ? "random" : "zero", flops / (best * 1e-3) * 1e-12, flops / (worst * 1e-3) * 1e-12);
    (void)hipFree(out);
}

int main() {
    for (int dense = 0; dense < 2; ++dense) {
        run<0>("f16 32x32x16 (4 accumulators)", dense);
        run<1>("f16 16x16x32 (16 accumulators)", dense);
    }
    run<0>("f16 32x32x16 (4 accumulators)", 1);
    run<1>("f16 16x16x32 (16 accumulators)", 1);
    return 0;
}
