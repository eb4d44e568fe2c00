// Issue rate of fp32 VALU instructions on gfx950: scalar v_fma_f32 vs packed v_pk_fma_f32, one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/probes/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void rate(float* out, long long* cyc, int iters) {
    float a = out[threadIdx.x], b = out[threadIdx.x + 1];
    f2 x[8];
    for (int k = 0; k < 8; ++k) x[k] = f2{a + k, b - k};
    const f2 m = {a, a}, c = {b, b};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(m), "v"(c));
            else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k].x) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k].y) : "v"(a), "v"(b)); }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += x[k].x + x[k].y;
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1 << 20); (void)hipMemset(out, 0, 1 << 20); (void)hipMalloc(&cyc, 8);
    const int iters = 4096;
    for (int pk = 0; pk < 2; ++pk)
        for (int threads : {64, 256, 512, 1024}) {
            long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (pk) hipLaunchKernelGGL(rate<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
                else hipLaunchKernelGGL(rate<0>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            }
            const double fmas = (double)iters * 16;            // scalar-FMA equivalents per lane
            printf("%s threads %4d (waves/SIMD %.2f): %lld ticks, %.2f ticks per fp32 FMA per wave (s_memtime ticks at 100 MHz x clock ratio; compare rows)\n",
                   pk ? "v_pk_fma_f32" : "v_fma_f32   ", threads, threads / 256.0, h, (double)h / fmas);
        }
    return 0;
}
