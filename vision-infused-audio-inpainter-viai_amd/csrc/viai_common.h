// Shared device/host helpers for the VIAI gfx950 kernels.
// All activations are NHWC fp32 ([N][H][W][C], C contiguous); see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <tuple>
#include <utility>
#include <stdint.h>

#define VIAI_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#include "../../include/viai_hip.h"   // activation codes, ABI structs

#define VIAI_MAX_TAPS 16

// Convolution-like gather described by a tap table.  One launch computes the
// output pixels of one sub-lattice:  out(y, x) = (oy*ly + ay, ox*lx + ax),
// oy < SH, ox < SW, reading in(oy*my + dy[t], ox*mx + dx[t]) for every tap t.
//   forward Conv2d stride s, pad p      : l=1 a=0 m=s  d[t] = r - p
//   forward ConvTranspose2d stride 1    : l=1 a=0 m=1  d[t] = p - r
//   dgrad of Conv2d, parity class a     : l=s a=a m=1  d[t] = (a + p - r)/s, r = (a+p) mod s ...
//   dgrad of ConvTranspose2d stride 1   : l=1 a=0 m=1  d[t] = r - p
struct ConvGeom {
    int N, IH, IW;        // gathered tensor
    int OH, OW;           // full extent of the produced tensor
    int SH, SW;           // sub-lattice extent of this launch
    int ly, lx, ay, ax;
    int my, mx;
    int ntaps;            // taps used by this launch
    int run;              // 1: "row-run" mode for 1 < Cin <= 4 (ResNet conv1 7x7): a tap is a whole kernel ROW and
                          //    its 32 K-values are 8 consecutive pixels x 4 (zero-padded) channels; lane quad q <-> pixel +q
    int wtaps;            // tap slots per packed-weight row
    // 32-bit entries: the kernels index these with a wave-uniform tap id, which must lower to a scalar
    // s_load (byte-sized entries became per-lane global loads + vmcnt(0) stalls in the K loop)
    int dy[VIAI_MAX_TAPS], dx[VIAI_MAX_TAPS], ws[VIAI_MAX_TAPS];
};

// none / ReLU / LeakyReLU are ONE expression, max(v, 0) + ns min(v, 0) with ns = 1 / 0 / slope (the same values bit for bit: the fma rounds
// slope * v once, and 0 + (-0) = +0): with `act` a kernel argument the three-way branch used to sit in front of every element of the
// streaming kernels' unrolled loops; now the compiler hoists one scalar select and keeps a single uniform branch for the sigmoid
__device__ __forceinline__ float viai_act_ns(int act, float slope) { return act == VIAI_ACT_RELU ? 0.f : (act == VIAI_ACT_LRELU ? slope : 1.f); }
__device__ __forceinline__ float viai_act(float v, int act, float slope) {
    if (act == VIAI_ACT_SIGMOID) return 1.f / (1.f + expf(-v));   // accurate exp: BCE divides by p(1-p)
    // fminf / fmaxf return the non-NaN operand, so the sum alone would turn a NaN into 0 and a diverged network would report finite
    // losses; torch (and the reference) propagate it: one compare + select per element, the finite values are unchanged bit for bit
    const float r = fmaf(viai_act_ns(act, slope), fminf(v, 0.f), fmaxf(v, 0.f));
    return v != v ? v : r;
}
// d act / d pre for the piecewise-linear activations (sigmoid: see the callers)
__device__ __forceinline__ float viai_act_grad_pl(float pre, int act, float slope) { return pre > 0.f ? 1.f : viai_act_ns(act, slope); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The same sum (different association) without the LDS crossbar: `__shfl_xor` compiles to ds_bpermute_b32, ~100+ cycles each and six
// of them in a dependent chain per value; four DPP adds (quad swaps, half-row and row mirrors: ALU latency) leave every lane with its
// 16-lane row sum, four v_readlane fetch the row sums.  For latency-bound kernels that reduce several values per wave.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

// fp64 flavour: the two halves move through DPP separately
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double readlane_d(double x, int lane) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_sum_d_dpp(double v) {
    v += dpp_mov_d<0xB1>(v);
    v += dpp_mov_d<0x4E>(v);
    v += dpp_mov_d<0x141>(v);
    v += dpp_mov_d<0x140>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// max |.| of a tensor into a device float (zero-initialised by the caller, or holding an earlier maximum): wave shuffle, LDS, at most one
// conditional atomicMax per block on the float bits (non-negative floats order like unsigned integers; max is order-independent, so the
// result is deterministic).  Blocks of 64 .. 1024 threads; every thread of the block must call it.
// The atomics of one launch all hit ONE address and are executed at the memory side one after the other (~10 ns each): a launch of 2048
// blocks that all finish together -- and therefore all read the slot while it still holds 0 -- spends 10 - 20 us in them after its last
// store (tools/probes/bn_fresh.py: the 16.8 MB BatchNorm apply pass 10 -> 30 us).  Kernels that end in this call run as FEW, FAT blocks
// (one 1024-thread block per CU) for that reason.
__device__ __forceinline__ void block_absmax_to(float* amax, float mx) {
    __shared__ float viai_wmax[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) viai_wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)(blockDim.x >> 6);
        for (int w = 1; w < nw; ++w) mx = fmaxf(mx, viai_wmax[w]);
        // plain read first: thousands of blocks hammering one address with atomics measured +0.9 ms per step
        if (mx > 0.f && __float_as_uint(mx) > __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(amax)))
            atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(mx));
    }
}

// XCD-aware bijective remap of a linear workgroup id: consecutive logical tiles
// land on the same XCD (block b is observed to run on XCD b % 8), so
// neighbouring tiles share that XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int q = nwg / nx, r = nwg % nx;
    int xcd = bid % nx, idx = bid / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Launch + error capture.  hipGetLastError() is a sticky per-thread slot that other HIP users in the
// process (torch's own probing calls) may have left set, so it is cleared before every launch and only
// errors raised by OUR launches are accumulated (first error wins) and reported by the entry point.
static thread_local int viai_err_acc = 0;
// Launch plans (plan.hip): while the log is on, every launch notes (kernel, launch geometry, argument bytes, stream) so that a
// plan built from the stream capture of the same launches can put each node back on the stream it was issued to (a node is
// matched to its note by kernel + geometry + argument bytes, whatever order the graph API returns the nodes in).
extern int viai_plan_log_on;
void viai_plan_note(const void* func, void* stream, dim3 grid, dim3 block, const unsigned char* blob, const unsigned* sizes, int nargs);
// One launch: the argument expressions are evaluated ONCE, converted to the kernel's parameter types, and the same packed values go to
// hipLaunchKernel and (while a plan log is on) into the launch note.
template <class... P, class... A, size_t... I>
static inline void viai_launch_impl(void (*f)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, std::index_sequence<I...>, A&&... a) {
    static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
    std::tuple<std::remove_cv_t<P>...> vals{static_cast<P>(a)...};
    void* ptrs[sizeof...(P) + 1] = {(void*)&std::get<I>(vals)..., nullptr};
    (void)hipLaunchKernel(reinterpret_cast<const void*>(f), grid, block, ptrs, lds, st);
    if (viai_plan_log_on) {
        unsigned char blob[(sizeof(P) + ... + 16)];
        unsigned sizes[sizeof...(P) + 1];
        int n = 0;
        size_t off = 0;
        auto put = [&](const void* p, size_t bytes) { __builtin_memcpy(blob + off, p, bytes); off += bytes; sizes[n++] = (unsigned)bytes; };
        (put(&std::get<I>(vals), sizeof(P)), ...);
        viai_plan_note(reinterpret_cast<const void*>(f), (void*)st, grid, block, blob, sizes, n);
    }
}
template <class... P, class... A>
static inline void viai_launch_once(void (*f)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A&&... a) {
    viai_launch_impl(f, grid, block, lds, st, std::index_sequence_for<P...>{}, static_cast<A&&>(a)...);
}
#define VIAI_LAUNCH(...)                                         \
    do {                                                         \
        (void)hipGetLastError();                                 \
        viai_launch_once(__VA_ARGS__);                           \
        int viai_e_ = (int)hipGetLastError();                    \
        if (viai_e_ != 0 && viai_err_acc == 0) viai_err_acc = viai_e_; \
    } while (0)
static inline int viai_launch_status() { int e = viai_err_acc; viai_err_acc = 0; return e; }
