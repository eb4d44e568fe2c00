// Sustained-rate probe for v_mfma_f32_32x32x16_bf16 on gfx950 (no memory traffic): what the matrix pipes deliver
// under continuous load, i.e. the practical ceiling the conv kernels are priced against in DESIGN.md section 3.3.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, int dense) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    // operands: dense=0 -> small constants (few toggling bits), dense=1 -> pseudo-random full-mantissa values
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int e = 0; e < 8; ++e) {
        h = h * 1664525u + 1013904223u;
        float ra = dense ? ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) : (float)(threadIdx.x & 3);
        h = h * 1664525u + 1013904223u;
        float rb = dense ? ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.01f : 1.0f;
        a[e] = (__bf16)ra; b[e] = (__bf16)rb;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same MFMA stream fed from LDS the way the conv kernels feed it: 12 ds_read_b128 operand fetches per 48 MFMAs
// (the frag kernel's ratio), random bf16 data, no global traffic and no barriers inside the loop
__global__ __launch_bounds__(256) void probe_lds(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[48 * 1024];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 256) {
        h = h * 1664525u + 1013904223u;
        float r0 = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        h = h * 1664525u + 1013904223u;
        float r1 = ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.01f;
        __bf16 b0 = (__bf16)r0, b1 = (__bf16)r1;
        unsigned short u0 = __builtin_bit_cast(unsigned short, b0), u1 = __builtin_bit_cast(unsigned short, b1);
        reinterpret_cast<unsigned*>(sm)[i] = (unsigned)u0 | ((unsigned)u1 << 16);
    }
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int lane = threadIdx.x & 63;
    const unsigned char* base = sm + (lane & 31) * 80 + (lane >> 5) * 16;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* p = base + ((it & 7) * 4096);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    a[i][q] = *reinterpret_cast<const bf16x8*>(p + (i * 3 + q) * 2560 + ks * 32);
                    b[i][q] = *reinterpret_cast<const bf16x8*>(p + 16384 + (i * 3 + q) * 2560 + ks * 32);
                }
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][q % 3], b[j][(q + 1) % 3], acc[i * 2 + j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

void run_lds(int blocks_per_cu) {
    const int iters = 800, blocks = 256 * blocks_per_cu;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe_lds<<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    float best = 1e30f, worst = 0.f;
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0);
        probe_lds<<<blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        if (ms > worst) worst = ms;
    }
    double flops = (double)blocks * 4 * iters * 48 * 32768.0;
    printf("bf16 32x32x16 fed from LDS (12 ds_read_b128 per 48 MFMAs, random data): %d blocks/CU: best %.1f TFLOP/s, sustained %.1f TFLOP/s\n",
           blocks_per_cu, flops / (best * 1e-3) * 1e-12, flops / (worst * 1e-3) * 1e-12);
    hipFree(out);
}

template <int NACC>
void run(const char* name, int blocks_per_cu, int dense = 0) {
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC><<<blocks, 256>>>(out, 10, dense);
    hipDeviceSynchronize();
    float best = 1e30f, worst = 0.f;
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0);
        probe<NACC><<<blocks, 256>>>(out, iters, dense);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        if (ms > worst) worst = ms;
    }
    double flops = (double)blocks * 4 * iters * 8 * NACC * 32768.0;
    printf("%s (%s operands): %d blocks/CU, %d chains/wave: best %.1f TFLOP/s, sustained (slowest of 8) %.1f TFLOP/s\n", name, dense ? "random" : "constant", blocks_per_cu, NACC,
           flops / (best * 1e-3) * 1e-12, flops / (worst * 1e-3) * 1e-12);
    hipFree(out);
}

int main() {
    run<1>("bf16 32x32x16", 1);
    run<2>("bf16 32x32x16", 1);
    run<4>("bf16 32x32x16", 1);
    run<4>("bf16 32x32x16", 2);
    run<2>("bf16 32x32x16", 2);
    run<1>("bf16 32x32x16", 2);
    run<1>("bf16 32x32x16", 4);
    run<4>("bf16 32x32x16", 2, 1);
    run<4>("bf16 32x32x16", 2, 1);
    run<1>("bf16 32x32x16", 4, 1);
    run_lds(2);
    run_lds(2);
    run_lds(1);
    return 0;
}
