// probe: what does s_memtime count?  (stamps of tools/probes/halo_dma_bench.hip).  A wave issues 4096 dependent v_add_f32 (>= 4 shader cycles each... the
// exact count does not matter): the same work is timed by s_memtime, s_memrealtime (100 MHz) under an idle chip and again right after a 2 ms busy kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* o, int n) { float x = threadIdx.x; for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f; o[threadIdx.x + blockIdx.x * blockDim.x] = x; }
__global__ void probe(unsigned long long* out, float* o) {
    float x = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 64
    for (int i = 0; i < 4096; ++i) x = x * 1.0001f + 0.5f;
    asm volatile("" : "+v"(x));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    o[threadIdx.x] = x;
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
int main() {
    unsigned long long* d; float* o; hipMalloc(&d, 16); hipMalloc(&o, 4 * 1024 * 256 * 64);
    unsigned long long h[2];
    for (int rep = 0; rep < 3; ++rep) {
        hipDeviceSynchronize();
        probe<<<1, 64>>>(d, o); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("idle chip : s_memtime delta %llu, s_memrealtime delta %llu (x10 ns)\n", h[0], h[1]);
        spin<<<1024, 256>>>(o, 200000); probe<<<1, 64>>>(d, o); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("after busy: s_memtime delta %llu, s_memrealtime delta %llu (x10 ns)\n", h[0], h[1]);
    }
    return 0;
}
