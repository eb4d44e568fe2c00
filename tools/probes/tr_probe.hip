// Probe of ds_read_b64_tr_b16 (gfx950): which 16-bit element lands in which lane / slot.
// LDS holds l[i] = i; lane L supplies the byte address 8*L (its "own" four consecutive elements 4L..4L+3).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* o) {
    __shared__ __attribute__((aligned(16))) short l[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) l[i] = (short)i;
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(l + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) o[threadIdx.x * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int L = 0; L < 64; ++L) printf("lane %2d: %4d %4d %4d %4d\n", L, h[L * 4], h[L * 4 + 1], h[L * 4 + 2], h[L * 4 + 3]);
    return 0;
}
