// Does v_mfma_f32_32x32x2_f32 round its accumulation to nearest?  One wave multiplies A (32 x K) by B (K x 32) through the MFMA chain the
// exact-fp32 conv kernels use; the host compares with fp64 and with an fmaf chain in the same order: mean SIGNED error (a bias shows as a
// mean many sigma from zero), rms error, and how many results equal the fmaf chain bit for bit.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_bias.hip -o /tmp/mfma_f32_bias && /tmp/mfma_f32_bias
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc; for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], B[(k0 + h) * 32 + r], acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) { const int row = (e & 3) + 8 * (e >> 2) + 4 * h; C[row * 32 + r] = acc[e]; }
}
int main() {
    const int K = 4608;
    std::vector<float> A(32 * K), B(K * 32), C(1024);
    srand(1);
    for (int mode = 0; mode < 2; ++mode) {
        for (auto& v : A) v = (float)rand() / RAND_MAX * (mode ? 1.f : 2.f) - (mode ? 0.f : 1.f);      // mode 1: all-positive operands (sums grow)
        for (auto& v : B) v = (float)rand() / RAND_MAX * (mode ? 1.f : 2.f) - (mode ? 0.f : 1.f);
        float *dA, *dB, *dC;
        (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 4096);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, K);
        (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        double me = 0, se = 0, mf = 0, sf = 0, scale = 0; int same = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double t = 0; float f = 0.f;
            for (int kk = 0; kk < K; ++kk) { t += (double)A[i * K + kk] * B[kk * 32 + j]; f = fmaf(A[i * K + kk], B[kk * 32 + j], f); }
            const double e = C[i * 32 + j] - t, ef = f - t;
            me += e; se += e * e; mf += ef; sf += ef * ef; scale += t * t; same += (C[i * 32 + j] == f);
        }
        const double n = 1024, rms = sqrt(se / n), rmsf = sqrt(sf / n);
        printf("%s operands, K = %d: |C| rms %.3g\n  mfma : mean err %+.3e (%.1f sigma of the mean), rms err %.3e\n  fmaf : mean err %+.3e (%.1f sigma), rms err %.3e\n  bit-identical to the fmaf chain: %d / 1024\n",
               mode ? "positive" : "signed", K, sqrt(scale / n), me / n, fabs(me / n) / (rms / sqrt(n)), rms, mf / n, fabs(mf / n) / (rmsf / sqrt(n)), rmsf, same);
    }
    return 0;
}
