#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// probe: buffer_load_dwordx4 ... lds  -- (1) does an out-of-range lane write ZEROS to LDS?  (2) lane-linear placement
__global__ void k(const float* src, int nbytes, float* out, int shift) {
    __shared__ __attribute__((aligned(16))) float lds[2][256];
    for (int i = threadIdx.x; i < 512; i += 64) (&lds[0][0])[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    int voff = (threadIdx.x + shift) * 16;
    if (threadIdx.x == 5) voff |= 0x7fffffff;            // forced out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&lds[1][0], 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = (&lds[0][0])[i];
}
int main() {
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = i + 1;
    float *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 2048); hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    // buffer covers only 60 lanes x 16 B = 960 bytes: lanes 60..63 out of range too
    k<<<1, 64>>>(d, 960, o, 0);
    std::vector<float> r(512); hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    printf("lds[0] untouched: %g %g\n", r[0], r[255]);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %g %g %g %g\n", l, r[256 + 4 * l], r[256 + 4 * l + 1], r[256 + 4 * l + 2], r[256 + 4 * l + 3]);
    return 0;
}
