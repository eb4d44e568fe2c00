// WaveNet incremental synthesis as ONE persistent launch: a weight-stationary pipeline over the chip (gfx950, round 6).
// Reference: wavenet_vocoder/wavenet.py:237-364 (incremental_forward), conv.py:17-46 (ring-buffer convolution), modules.py:162-210
// (ResidualConv1dGLU.incremental_forward), mixture.py:117-153 (sampler).
//
// The chain form (wavenet.hip) runs a time step as 27 dependent launches that each pull the layer's 4 MB of weights through the whole chip:
// 142.7 us per time step, of which the weights' memory time is 12 us -- the step is a chain of hand-offs, not a stream.  A time step of ONE
// stream cannot be cut shorter than its 27 dependent stages, but the 8 streams of the benchmark are independent, so the stages can work on
// different streams at the same moment.  Here every stage OWNS compute units and its weights never move again:
//
//   stage l (l < 24): 10 compute units hold layer l's fused rows (the extended gate rows [Wc^0 | Wc^1 | c | r Wc^2 | r Wc^2 Wo_prev] of the chain
//                     form, and the out / skip rows of layer l - 1) -- 182 registers per lane of the past-tap / current-tap columns, 136 KB of
//                     LDS for the z columns and the out / skip rows; CU j owns gate pairs h in [26 j, 26 j + 26), residual rows
//                     [52 j, 52 j + 52), skip rows [26 j, 26 j + 26);
//   stage 24 / 25:    4 compute units each: the last layer's skip rows (+ ReLU), the head's first 1x1 (+ ReLU);
//   stage 26:         1 compute unit: the head's second 1x1 and the mixture-of-logistics sample, which is stage 0's next input.
//
// A TOKEN is one stream's state at one time step on its way from stage to stage: 1024 values (x_l(t) 512 | z_l(t) 256 | running skip sum 256)
// published as 8-byte {tag = t + 1, value} granules with write-through stores (MI355X guide, Guideline 16 form R2: the data is the flag; no
// fence, no separate flag).  A stage's token slots form a ring over time (2 d + 2 slots per stream), so the ring IS the layer's past-tap
// buffer of conv.py: x_l(t - d) and x_l(t - 2 d) are read back from it, tags checked.  The eight streams go round the 27 stages one behind
// the other; a stage computes what does not depend on the arriving token (two of the three taps, the conditioning) while it waits.
//
// Every wait is bounded (a timeout raises a device flag that every other wait polls) and nothing depends on where a block runs; all 249
// blocks must be resident at once, which the host checks (256 compute units, > 80 KB of LDS per block = one block per CU).
// Arithmetic: fp32 FMA, fixed summation order (deterministic); same folded weights as the chain form.
#include "viai_common.h"
#include "viai_internal.h"

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int PC = 512, PH = 256, PS = 256, PCIN = 80;
constexpr int NL = 24, NCU = 10, NHS = 4, NH1 = 4;
constexpr int HSL = 26, XSL = 52, SSL = 26;           // per-CU slices of the gate pairs / residual rows / skip rows
constexpr int GW = 7, BW = 10;                        // row slots per wave: gate (56 slots, 52 used), out + skip (80 slots, 78 used)
constexpr int NW = 8, NT = 512;
constexpr int KPRE = 1024;                            // [x(t - 2d) 512 | x(t - d) 512]: 16 columns per lane
constexpr int PRE_M = 4;                              // 4 x (4 columns per lane)
constexpr int NREG = GW * 16 + GW * 8;                // 168 weight registers per lane
constexpr int CROWS = 64;                             // conditioning rows: 8 lanes per row (streamed from L2 while the stage waits)
constexpr int TOK = 1024, TOK_X = 0, TOK_Z = 512, TOK_S = 768;
constexpr int NSTAGE = NL + 3;
constexpr int LDS_ROWS = NW * GW + NW * BW;           // 136 rows of 256 floats
constexpr unsigned SPIN_LIMIT = 400000u;              // ~0.3 - 0.5 s of polling: a stage that starves this long has lost its producer

struct WnPipe {
    const float* wreg;        // [NL][NCU][NW][NREG][64]
    const float* wcond;       // [NL][NCU][CROWS][80]: conditioning columns of the gate row slots (slots >= 56: zero)
    const float* wlds;        // [NL][NCU][LDS_ROWS][256]
    const float* bias;        // [NL][NCU][LDS_ROWS]
    const float* head_w;      // [NHS*64 + NH1*64 + 32][256]
    const float* head_b;      // [NHS*64 + NH1*64 + 32]
    const float* w_first; const float* b_first;
    const float* cond;        // [B][T][80]
    const float* test_inputs; // [B][n_test]
    const float* u1; const float* u2;
    float* out; float* yhat_dbg;
    u64* tok;                 // all rings
    unsigned* err;            // [0] flag, [1..3] where
    long tok_off[NSTAGE];     // granule offset of a stage's rings: [B][rl][TOK]
    int rl[NSTAGE];
    int dil[NL];
    int B, T, n_test, t0, t1, out_ch;
    float log_scale_min;
};

__device__ __forceinline__ u64 gload(const u64* p) { return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gstore(u64* p, unsigned tag, float v) {
    __hip_atomic_store((gu64*)p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned eload(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ u64* slot_of(const WnPipe& a, int st, int s, int t) {
    return a.tok + a.tok_off[st] + ((long)s * a.rl[st] + (t % a.rl[st])) * TOK;
}

// where a block sits: blocks b and b + 8 usually share an XCD (observed, not relied on), so three consecutive layers get the 30 first
// blocks of an XCD and the head the spare ones
struct Role { int st, j; };
__device__ __forceinline__ Role role_of(int b) {
    const int x = b & 7, i = b >> 3;                  // "XCD", index within it (0 .. 31)
    if (i < 30) return Role{3 * x + i / NCU, i % NCU};
    const int sp = x * 2 + (i - 30);                  // 16 spare blocks
    if (sp >= 8 && sp < 8 + NHS) return Role{NL, sp - 8};            // XCD 4, 5
    if (sp >= 12 && sp < 12 + NH1) return Role{NL + 1, sp - 12};     // XCD 6, 7
    if (sp == 0) return Role{NL + 2, 0};                             // XCD 0 (next to layer 0)
    return Role{-1, 0};
}

// bounded wait: `ok` = this thread's granules carry the tag; returns false when the block must give up
__device__ __forceinline__ bool poll_fail(const WnPipe& a, unsigned& spins, int st, int s, int t) {
    ++spins;
    int bad = 0;
    if (threadIdx.x == 0) {
        if (spins > SPIN_LIMIT) {
            bad = 1;
            if (atomicCAS(a.err, 0u, 1u) == 0u) { a.err[1] = (unsigned)st; a.err[2] = (unsigned)s; a.err[3] = (unsigned)t; }
        } else if ((spins & 63u) == 0u && eload(a.err) != 0u) bad = 1;
    }
    if (__syncthreads_or(bad)) return true;
    __builtin_amdgcn_s_sleep(1);
    return false;
}

__device__ __forceinline__ float dot4(const f32x4 w, const f32x4 x, float acc) {
    acc = fmaf(w[0], x[0], acc); acc = fmaf(w[1], x[1], acc); acc = fmaf(w[2], x[2], acc); return fmaf(w[3], x[3], acc);
}

// ------------------------------------------------------------------------------------------------------------------ a layer stage
__device__ void layer_stage(const WnPipe& a, const int l, const int j, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wz = lds;                                  // [56][256] gate rows, z columns
    float* wb = wz + NW * GW * 256;                   // [80][256] out / skip rows
    float* xpre = wb + NW * BW * 256;                 // [1152]
    float* xcur = xpre + KPRE;                        // [512]
    float* zin = xcur + PC;                           // [256]
    float* skin = zin + PH;                           // [32]
    float* gsum = skin + 32;                          // [56]
    float* bres = gsum + 64;                          // [80]
    float* bia = bres + 80;                           // [136]
    float* cnd = bia + LDS_ROWS;                      // [80] c_t
    float* cpart = cnd + PCIN;                        // [64] conditioning contribution of each gate row slot
    // ---- weights: never move again
    float w[NREG];
    {
        const float* src = a.wreg + (((size_t)(l * NCU + j) * NW + wave) * NREG) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NREG; ++i) w[i] = src[(size_t)i * 64];
        const f32x4* ls = reinterpret_cast<const f32x4*>(a.wlds + (size_t)(l * NCU + j) * LDS_ROWS * 256);
        for (int i = tid; i < LDS_ROWS * 64; i += NT) reinterpret_cast<f32x4*>(lds)[i] = ls[i];
        if (tid < LDS_ROWS) bia[tid] = a.bias[(size_t)(l * NCU + j) * LDS_ROWS + tid];
    }
    const float* wcrow = a.wcond + ((size_t)(l * NCU + j) * CROWS + (tid >> 3)) * PCIN + (tid & 7) * 10;
    __syncthreads();
    const int d = a.dil[l];
    const float r5 = 0.70710678118654752f;
    for (int t = a.t0; t < a.t1; ++t) {
        for (int s = 0; s < a.B; ++s) {
            // ---- what does not depend on the arriving token: the two past taps (this stage's own ring) and the conditioning
            {
                float v0 = 0.f, v1 = 0.f;
                bool ok = true;
                if (t - 2 * d >= 0) { const u64 g = gload(slot_of(a, l, s, t - 2 * d) + TOK_X + tid); ok &= (unsigned)(g >> 32) == (unsigned)(t - 2 * d + 1); v0 = __uint_as_float((unsigned)g); }
                if (t - d >= 0) { const u64 g = gload(slot_of(a, l, s, t - d) + TOK_X + tid); ok &= (unsigned)(g >> 32) == (unsigned)(t - d + 1); v1 = __uint_as_float((unsigned)g); }
                xpre[tid] = v0; xpre[PC + tid] = v1;
                if (tid < PCIN) cnd[tid] = a.cond[((size_t)s * a.T + t) * PCIN + tid];
                if (!ok && atomicCAS(a.err, 0u, 2u) == 0u) { a.err[1] = (unsigned)l; a.err[2] = (unsigned)s; a.err[3] = (unsigned)t; }   // a past tap that is not there: protocol defect
            }
            __syncthreads();
            float acc[GW];
#pragma unroll
            for (int i = 0; i < GW; ++i) acc[i] = 0.f;
#pragma unroll
            for (int m = 0; m < PRE_M; ++m) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xpre + 256 * m + 4 * lane);
#pragma unroll
                for (int i = 0; i < GW; ++i) {
                    const int r = i * 16 + 4 * m;
                    acc[i] = fmaf(w[r], xv[0], acc[i]); acc[i] = fmaf(w[r + 1], xv[1], acc[i]); acc[i] = fmaf(w[r + 2], xv[2], acc[i]); acc[i] = fmaf(w[r + 3], xv[3], acc[i]);
                }
            }
            {   // conditioning columns (modules.py:189-193 conv1x1c): row slot tid / 8, ten columns per lane, weights from L2 -- nothing waits for this
                float cs = 0.f;
#pragma unroll
                for (int k = 0; k < 10; ++k) cs = fmaf(wcrow[k], cnd[(tid & 7) * 10 + k], cs);
                cs += __shfl_xor(cs, 1, 64); cs += __shfl_xor(cs, 2, 64); cs += __shfl_xor(cs, 4, 64);
                if ((tid & 7) == 0) cpart[tid >> 3] = cs;
            }
            // ---- the token: x_{l-1}(t) | z_{l-1}(t) | skip sum (own rows) from stage l - 1; stage 0: the previous sample
            unsigned spins = 0;
            if (l == 0) {
                float cur = 0.f;
                if (t < a.n_test) cur = a.test_inputs[(size_t)s * a.n_test + t];
                else if (t > 0) {
                    const u64* p = slot_of(a, NL + 2, s, t - 1);
                    for (;;) {
                        const u64 g = gload(p);
                        const bool ok = (unsigned)(g >> 32) == (unsigned)t;
                        cur = __uint_as_float((unsigned)g);
                        if (__syncthreads_and(ok)) break;
                        if (poll_fail(a, spins, l, s, t)) return;
                    }
                }
                xcur[tid] = fmaf(cur, a.w_first[tid], a.b_first[tid]);          // wavenet.py:118 first_conv
                __syncthreads();
            } else {
                const u64* p = slot_of(a, l - 1, s, t);
                const bool has2 = tid < PH + SSL;
                const int i2 = tid < PH ? TOK_Z + tid : TOK_S + min(SSL * j + (tid - PH), PS - 1);
                for (;;) {
                    const u64 g0 = gload(p + TOK_X + tid);
                    bool ok = (unsigned)(g0 >> 32) == (unsigned)(t + 1);
                    xcur[tid] = __uint_as_float((unsigned)g0);
                    if (has2) {
                        const u64 g1 = gload(p + i2);
                        ok &= (unsigned)(g1 >> 32) == (unsigned)(t + 1);
                        if (tid < PH) zin[tid] = __uint_as_float((unsigned)g1); else skin[tid - PH] = __uint_as_float((unsigned)g1);
                    }
                    if (__syncthreads_and(ok)) break;
                    if (poll_fail(a, spins, l, s, t)) return;
                }
            }
            // ---- current tap (registers), z columns and the out / skip rows (LDS)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xcur + 256 * m + 4 * lane);
#pragma unroll
                for (int i = 0; i < GW; ++i) {
                    const int r = GW * 16 + i * 8 + 4 * m;
                    acc[i] = fmaf(w[r], xv[0], acc[i]); acc[i] = fmaf(w[r + 1], xv[1], acc[i]); acc[i] = fmaf(w[r + 2], xv[2], acc[i]); acc[i] = fmaf(w[r + 3], xv[3], acc[i]);
                }
            }
            float bacc[BW];
            if (l > 0) {
                const f32x4 zv = *reinterpret_cast<const f32x4*>(zin + 4 * lane);
                // (two groups of five rows: with all ten in flight hipcc keeps 40 more registers live than the weights leave room for)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int i = 5 * h; i < 5 * h + 5; ++i) bacc[i] = dot4(*reinterpret_cast<const f32x4*>(wb + (wave * BW + i) * 256 + 4 * lane), zv, 0.f);
#pragma unroll
                    for (int i = 5 * h; i < 5 * h + 5; ++i) { const float v = wave_sum_dpp(bacc[i]); if (lane == 0) bres[wave * BW + i] = v; }
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int i = 0; i < GW; ++i) acc[i] = dot4(*reinterpret_cast<const f32x4*>(wz + (wave * GW + i) * 256 + 4 * lane), zv, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < GW; ++i) { const float v = wave_sum_dpp(acc[i]); if (lane == 0) gsum[wave * GW + i] = v; }
            __syncthreads();
            // ---- publish: z_l (26 pairs), x_l(t) (52 rows), the skip sum (26 rows)
            u64* q = slot_of(a, l, s, t);
            const unsigned tag = (unsigned)(t + 1);
            if (tid < HSL) {
                const int h = HSL * j + tid;
                if (h < PH) {
                    const float va = gsum[tid] + bia[tid] + cpart[tid], vg = gsum[HSL + tid] + bia[HSL + tid] + cpart[HSL + tid];
                    gstore(q + TOK_Z + h, tag, tanhf(va) * (1.f / (1.f + expf(-vg))));            // modules.py:201
                }
            } else if (tid >= 64 && tid < 64 + XSL) {
                const int k = tid - 64, c = XSL * j + k;
                if (c < PC) gstore(q + TOK_X + c, tag, l == 0 ? xcur[c] : (bres[k] + bia[NW * GW + k] + xcur[c]) * r5);          // modules.py:204-206
            } else if (tid >= 128 && tid < 128 + SSL) {
                const int k = tid - 128, si = SSL * j + k;
                if (si < PS) {
                    float v = 0.f;
                    if (l == 1) v = bres[XSL + k] + bia[NW * GW + XSL + k];
                    else if (l > 1) v = (skin[k] + bres[XSL + k] + bia[NW * GW + XSL + k]) * r5;          // wavenet.py:343-346
                    gstore(q + TOK_S + si, tag, v);
                }
            }
            // (no barrier: the next token's staging writes xpre only, and its first barrier orders everything after these reads)
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ head stages
// rows of a 256-wide 1x1 layer held in LDS; `mode` 0: v = relu((skin + W z + b) sqrt(.5)) (the last layer's skip rows), 1: v = relu(W x + b)
__device__ void dense_stage(const WnPipe& a, const int st, const int j, const int mode, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wr = lds;                  // [64][256]
    float* xin = wr + 64 * 256;       // [256]
    float* skin = xin + 256;          // [64]
    float* bia = skin + 64;           // [64]
    const int row0 = (mode == 0 ? 0 : NHS * 64) + 64 * j;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.head_w + (size_t)row0 * 256);
        for (int i = tid; i < 64 * 64; i += NT) reinterpret_cast<f32x4*>(wr)[i] = src[i];
        if (tid < 64) bia[tid] = a.head_b[row0 + tid];
    }
    __syncthreads();
    const float r5 = 0.70710678118654752f;
    const int in_off = mode == 0 ? TOK_Z : 0;
    for (int t = a.t0; t < a.t1; ++t)
        for (int s = 0; s < a.B; ++s) {
            const u64* p = slot_of(a, st - 1, s, t);
            unsigned spins = 0;
            const bool act = tid < 256 || (mode == 0 && tid < 256 + 64);
            const int idx = tid < 256 ? in_off + tid : TOK_S + 64 * j + (tid - 256);
            for (;;) {
                bool ok = true;
                if (act) {
                    const u64 g = gload(p + idx);
                    ok = (unsigned)(g >> 32) == (unsigned)(t + 1);
                    if (tid < 256) xin[tid] = __uint_as_float((unsigned)g); else skin[tid - 256] = __uint_as_float((unsigned)g);
                }
                if (__syncthreads_and(ok)) break;
                if (poll_fail(a, spins, st, s, t)) return;
            }
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xin + 4 * lane);
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = dot4(*reinterpret_cast<const f32x4*>(wr + (wave * 8 + i) * 256 + 4 * lane), xv, 0.f);
            u64* q = slot_of(a, st, s, t);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = wave_sum_dpp(acc[i]);
                if (lane == i) {
                    const int k = wave * 8 + i;
                    float y = v + bia[k];
                    if (mode == 0) y = (skin[k] + y) * r5;                                        // wavenet.py:343-346, then :349 ReLU
                    gstore(q + 64 * j + k, (unsigned)(t + 1), y > 0.f ? y : 0.f);
                }
            }
            __syncthreads();                          // xin / skin are rewritten by the next poll
        }
}

// the head's second 1x1 (30 rows) and the sampler (mixture.py:117-153 with injected uniforms)
__device__ void sample_stage(const WnPipe& a, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wr = lds;                  // [32][256]
    float* xin = wr + 32 * 256;       // [256]
    float* yo = xin + 256;            // [32]
    float* bia = yo + 32;             // [32]
    const int row0 = NHS * 64 + NH1 * 64;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.head_w + (size_t)row0 * 256);
        for (int i = tid; i < 32 * 64; i += NT) reinterpret_cast<f32x4*>(wr)[i] = src[i];
        if (tid < 32) bia[tid] = a.head_b[row0 + tid];
    }
    __syncthreads();
    const int OC = a.out_ch, K3 = OC / 3;
    for (int t = a.t0; t < a.t1; ++t)
        for (int s = 0; s < a.B; ++s) {
            const u64* p = slot_of(a, NL + 1, s, t);
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                if (tid < 256) { const u64 g = gload(p + tid); ok = (unsigned)(g >> 32) == (unsigned)(t + 1); xin[tid] = __uint_as_float((unsigned)g); }
                if (__syncthreads_and(ok)) break;
                if (poll_fail(a, spins, NL + 2, s, t)) return;
            }
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xin + 4 * lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = wave_sum_dpp(dot4(*reinterpret_cast<const f32x4*>(wr + (wave * 4 + i) * 256 + 4 * lane), xv, 0.f));
                if (lane == 0) yo[wave * 4 + i] = v + bia[wave * 4 + i];
            }
            __syncthreads();
            if (tid < OC && a.yhat_dbg) a.yhat_dbg[((size_t)s * a.T + t) * OC + tid] = yo[tid];
            if (tid == 0) {
                float best = -INFINITY; int arg = 0;
                for (int k = 0; k < K3; ++k) {
                    const float v = yo[k] - logf(-logf(a.u1[((size_t)s * a.T + t) * K3 + k]));
                    if (v > best) { best = v; arg = k; }
                }
                const float m = yo[K3 + arg], ls = fmaxf(yo[2 * K3 + arg], a.log_scale_min), u = a.u2[(size_t)s * a.T + t];
                float x = m + expf(ls) * (logf(u) - logf(1.f - u));
                x = fminf(fmaxf(x, -1.f), 1.f);
                a.out[(size_t)s * a.T + t] = x;
                gstore(slot_of(a, NL + 2, s, t), (unsigned)(t + 1), x);
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(NT) void wn_pipe_kernel(const WnPipe a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Role r = role_of(blockIdx.x);
    if (r.st < 0) return;
    if (r.st < NL) layer_stage(a, r.st, r.j, lds);
    else if (r.st == NL) dense_stage(a, NL, r.j, 0, lds);
    else if (r.st == NL + 1) dense_stage(a, NL + 1, r.j, 1, lds);
    else sample_stage(a, lds);
}

constexpr size_t PIPE_LDS_BYTES = (size_t)(LDS_ROWS * 256 + KPRE + PC + PH + 32 + 64 + 80 + LDS_ROWS + PCIN + 64) * sizeof(float);

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------------------
extern "C" int viai_wn_pipe_ok(const viai_wn_synth* s) {
    if (!s || s->C != PC || s->G != 2 * PH || s->S != PS || s->cin != PCIN || s->n_layers != NL || s->out_ch != 30 || s->out_ch % 3 != 0) return 0;
    if (s->B < 1 || s->B > 8 || s->cond == nullptr) return 0;
    for (int l = 0; l < NL; ++l) {
        if (s->layers[l].g_add != nullptr || s->layers[l].w_stage == nullptr || s->layers[l].w_c == nullptr) return 0;
        if (s->layers[l].dilation < 1 || s->layers[l].dilation > 4096) return 0;
    }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus >= 256 ? 1 : 0;
}

extern "C" long viai_wn_pipe_image_floats(int which) {
    switch (which) {
    case 5: return (long)NL * NCU * CROWS * PCIN;                // wcond
    case 0: return (long)NL * NCU * NW * NREG * 64;              // wreg
    case 1: return (long)NL * NCU * LDS_ROWS * 256;              // wlds
    case 2: return (long)NL * NCU * LDS_ROWS;                    // bias
    case 3: return (long)(NHS * 64 + NH1 * 64 + 32) * 256;       // head_w
    case 4: return (long)(NHS * 64 + NH1 * 64 + 32);             // head_b
    default: return 0;
    }
}

// granules (8 bytes each) of the token rings for B streams and the given dilations
extern "C" long viai_wn_pipe_token_granules(int B, const int* dil) {
    long n = 0;
    for (int l = 0; l < NL; ++l) n += (long)B * (2 * dil[l] + 2) * TOK;
    return n + 3L * B * 2 * TOK;
}

// time steps [t0, t0 + n) of every stream in ONE launch.  `tok` must be zero before the call with t0 == 0 and carried over between calls;
// err: 4 zeroed uint32 (err[0] != 0 after the call: 1 = a wait timed out at (stage, stream, t) = err[1..3], 2 = a past tap was missing).
extern "C" int viai_wn_pipe_run(const viai_wn_synth* s, const float* wreg, const float* wcond, const float* wlds, const float* bias, const float* head_w, const float* head_b,
                                void* tok, unsigned* err, int t0, int n_steps, void* stream) {
    if (!viai_wn_pipe_ok(s) || t0 < 0 || n_steps < 0 || t0 + n_steps > s->T) return (int)hipErrorInvalidValue;
    if (n_steps == 0) return 0;
    WnPipe a{};
    a.wreg = wreg; a.wcond = wcond; a.wlds = wlds; a.bias = bias; a.head_w = head_w; a.head_b = head_b;
    a.w_first = s->w_first; a.b_first = s->b_first; a.cond = s->cond; a.test_inputs = s->test_inputs; a.u1 = s->u1; a.u2 = s->u2;
    a.out = s->out; a.yhat_dbg = s->yhat_dbg; a.tok = (u64*)tok; a.err = err;
    long off = 0;
    for (int l = 0; l < NL; ++l) { a.dil[l] = s->layers[l].dilation; a.rl[l] = 2 * a.dil[l] + 2; a.tok_off[l] = off; off += (long)s->B * a.rl[l] * TOK; }
    for (int k = NL; k < NSTAGE; ++k) { a.rl[k] = 2; a.tok_off[k] = off; off += (long)s->B * 2 * TOK; }
    a.B = s->B; a.T = s->T; a.n_test = s->n_test; a.t0 = t0; a.t1 = t0 + n_steps; a.out_ch = s->out_ch; a.log_scale_min = s->log_scale_min;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wn_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIPE_LDS_BYTES) != hipSuccess) return (int)hipGetLastError();
        attr_set = true;
    }
    VIAI_LAUNCH(wn_pipe_kernel, dim3(256), dim3(NT), PIPE_LDS_BYTES, (hipStream_t)stream, a);
    return viai_launch_status();
}
