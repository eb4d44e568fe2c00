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
//                     form, and the out / skip rows of layer l - 1) -- 176 registers per lane of the past-tap / current-tap columns, 135 KB of
//                     LDS for the z columns and the out / skip rows, the conditioning columns streamed from L2 while the stage waits; CU j owns
//                     gate pairs h in [26 j, 26 j + 26), residual rows [52 j, 52 j + 52), skip rows [26 j, 26 j + 26);
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
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int PC = 512, PH = 256, PS = 256, PCIN = 80;
constexpr int NL = 24, NCU = 10, NHS = 4, NH1 = 4;
constexpr int HSL = 26, XSL = 52, SSL = 26;           // per-CU slices of the gate pairs / residual rows / skip rows
constexpr int NW = 8, NT = 512;
constexpr int KPRE = 1024;                            // [x(t - 2d) 512 | x(t - d) 512]: 16 columns per lane
constexpr int NREG = 13 * 8 + 13 * 4;                 // 156 weight registers per lane: 13 rows x (8 past-tap + 4 current-tap columns)
constexpr int PBS = 96;                               // row length of the out / skip partial sums
constexpr int GROWS = 52, BROWS = 78, LROW = 260;     // LDS rows (gate z columns / out + skip), padded row length: lane = row reads are conflict-free
constexpr int CROWS = 64;                             // conditioning rows: 8 lanes per row (streamed from L2 while the stage waits)
constexpr int TOK = 1024, TOK_X = 0, TOK_Z = 512, TOK_S = 768;
constexpr int NSTAGE = NL + 3;
constexpr int PIPE_MAXB = 32;                            // streams per launch (the benchmark's 8 leave a stage idle 60 % of the time: the revolution is latency)
constexpr unsigned SPIN_LIMIT = 400000u;              // ~0.3 - 0.5 s of polling: a stage that starves this long has lost its producer

struct WnPipe {
    const float* wreg;        // [NL][NCU][NW][NREG][64]
    const float* wcond;       // [NL][NCU][CROWS][80]: conditioning columns of the gate row slots (slots >= 56: zero)
    const float* wlds;        // [NL][NCU][GROWS + BROWS][LROW]
    const float* bias;        // [NL][NCU][136]: 56 gate row slots, 80 out / skip row slots
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
    u64* prof; int prof_t;    // debug aid (viai_wn_pipe_profile): wall-clock stamps of time step prof_t, [stage][stream][8]
};

// stamps of one time step on CU 0 of every stage: 0 = the wait for the token begins, 1 = token complete, 2 = results in LDS, 3 = publish stores issued;
// layer stages also 4 = residual rows done, 5 = past barrier 1, 6 = gate rows done, 7 = past barrier 2 (thread 0's view)
__device__ __forceinline__ void stamp(const WnPipe& a, int st, int j, int s, int t, int k) {
    if (a.prof != nullptr && t == a.prof_t && j == 0 && s < 8 && threadIdx.x == 0) {
        a.prof[((size_t)st * 8 + s) * 8 + k] = wall_clock64();
        a.prof[27 * 8 * 8 + ((size_t)st * 8 + s) * 8 + k] = (u64)clock64();          // shader cycles beside the 100 MHz wall clock: the clock the stage runs at
    }
}

#ifdef VIAI_WN_FINE_STAMPS
#define FINE_STAMP(k) stamp(a, l, j, s, t, k)
#else
#define FINE_STAMP(k) do { } while (0)
#endif

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

// the same for a wave that polls on its own (no block barrier per poll): on a timeout or a raised flag it marks the block's LDS abort word and leaves
// its loop; every wave then meets at the phase's block barrier, sees the word and returns
__device__ __forceinline__ bool wave_poll_fail(const WnPipe& a, unsigned& spins, int* abortf, int st, int s, int t) {
    ++spins;
    bool bad = false;
    if (spins > SPIN_LIMIT) {
        bad = true;
        if ((threadIdx.x & 63) == 0 && atomicCAS(a.err, 0u, 1u) == 0u) { a.err[1] = (unsigned)st; a.err[2] = (unsigned)s; a.err[3] = (unsigned)t; }
    } else if ((spins & 63u) == 0u && eload(a.err) != 0u) bad = true;                  // (one address: uniform over the wave)
    if (bad) *abortf = 1;
    return bad;
}

__device__ __forceinline__ float dot4(const f32x4 w, const f32x4 x, float acc) {
    acc = fmaf(w[0], x[0], acc); acc = fmaf(w[1], x[1], acc); acc = fmaf(w[2], x[2], acc); return fmaf(w[3], x[3], acc);
}

// ------------------------------------------------------------------------------------------------------------------ a layer stage
// Register image, no slot wasted: a wave owns 128 past-tap columns and 64 current-tap columns of ALL 52 gate rows; its lane (g, cg) = (lane / 16, lane % 16)
// holds rows 13 g .. 13 g + 12 x 8 past-tap columns (104 registers) and x 4 current-tap columns (52 registers).  A lane's 13 partial sums are reduced over
// the 16 lanes of its row group with four DPP adds each (no readlane, no LDS crossbar), the eight waves' sums meet in LDS, and the thread that publishes a
// row adds them in a fixed order.  The z columns and the out / skip rows live in LDS rows padded to 260 floats, lane = row, the waves split the columns.
// (History, tools/wn_pipe_stamps.py: 17 full-wave reductions per wave behind the token cost 2.0 us of a 4 us stage; a lane = row register layout needed
// 176 registers of weights and the compiler spilt addresses into scratch on the publish path: 1.3 us.)
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror
    return v;
}

// the lane id, computed where it is used (two VALU instructions, opaque to the compiler): kept live across the token loop, hipcc spills it -- and LDS
// addresses derived from it -- to scratch and reloads them on the critical path, each reload a memory round trip
__device__ __forceinline__ int lane_now() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

__device__ __forceinline__ void layer_stage(const WnPipe& a, const int l, const int j, float* lds) {
    const int tid = threadIdx.x;                                                   // (prologue only)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);           // a scalar
    float* wz = lds;                                  // [52][260] gate rows, z columns
    float* wb = wz + GROWS * LROW;                    // [78][260] out / skip rows
    float* xpre = wb + BROWS * LROW;                  // [1024]
    float* xcur = xpre + KPRE;                        // [512]
    float* zin = xcur + PC;                           // [256]
    float* skin = zin + PH;                           // [32] (unused)
    float* cpart = skin + 32;                         // [64] conditioning part of each gate row
    float* bia = cpart + 64;                          // [136]
    float* cnd = bia + 136;                           // [80] c_t
    float* pg = cnd + PCIN;                           // [8][52] the waves' partial sums of the gate rows (past taps + current tap + z columns)
    float* pb = pg + NW * GROWS;                      // [8][96] out / skip partials: residual rows at 0 .. 51, skip rows at 64 .. 89
    int* abortf = reinterpret_cast<int*>(pb + NW * PBS);   // a wave gave up waiting (timeout / error flag raised elsewhere)
    if (tid == 0) *abortf = 0;
    // ---- weights: never move again
    float w[NREG];
    {
        const float* src = a.wreg + (((size_t)(l * NCU + j) * NW + wave) * NREG) * 64 + (tid & 63);
#pragma unroll
        for (int i = 0; i < NREG; ++i) w[i] = src[(size_t)i * 64];
        const f32x4* ls = reinterpret_cast<const f32x4*>(a.wlds + (size_t)(l * NCU + j) * (GROWS + BROWS) * LROW);
        for (int i = tid; i < (GROWS + BROWS) * LROW / 4; i += NT) reinterpret_cast<f32x4*>(lds)[i] = ls[i];
        if (tid < 136) bia[tid] = a.bias[(size_t)(l * NCU + j) * 136 + tid];
    }
    __syncthreads();
    const int d = a.dil[l];
    const float r5 = 0.70710678118654752f;
    // what the loops need of the argument block, once (hipcc re-read the arrays and re-derived `t % ring length` with a 30-instruction division per use)
    const int rl = a.rl[l], B = a.B, T = a.T, n_test = a.n_test;
    u64* const ring = a.tok + a.tok_off[l];
    const u64* const prev = l > 0 ? a.tok + a.tok_off[l - 1] : a.tok + a.tok_off[NL + 2];
    const int prl = l > 0 ? a.rl[l - 1] : 2;
    for (int t = a.t0; t < a.t1; ++t) {
        const int m0 = t % rl, m1 = t - d >= 0 ? (t - d) % rl : 0, m2 = t - 2 * d >= 0 ? (t - 2 * d) % rl : 0;
        const int mp = l > 0 ? t % prl : (t + 1) % 2;                          // stage 0 reads the sample of t - 1
        for (int s = 0; s < B; ++s) {
            // ---- what does not depend on the arriving token: the two past taps (this stage's own ring) and the conditioning.  A wave stages exactly the
            // 128 past-tap columns it multiplies (waves 0 - 3: x(t - 2d), waves 4 - 7: x(t - d)): no block barrier
            {
                const int dd = wave < 4 ? 2 * d : d, mm = wave < 4 ? m2 : m1;
                const int lane = lane_now();
                float v0 = 0.f, v1 = 0.f;
                if (t - dd >= 0) {
                    // The columns come from the stage's nine sibling blocks.  Nothing downstream orders a sibling's x_l(t - d) before this block's step t
                    // when d = 1 and there is one stream (the sample of t - 2 proves x_l(t - 2) only; at t = 1 a sibling may still be loading its weights),
                    // so this is a wait like any other: in the common case the first load carries the tag.  A sibling cannot lap the slot -- it is at
                    // most one time step ahead of this block, whose columns the next stage needs before the next sample exists.
                    const u64* pr = ring + ((long)s * rl + mm) * TOK + TOK_X + 128 * (wave & 3) + lane;
                    const unsigned want = (unsigned)(t - dd + 1);
                    for (unsigned spins = 0;;) {
                        const u64 ga_ = gload(pr), gb_ = gload(pr + 64);
                        v0 = __uint_as_float((unsigned)ga_); v1 = __uint_as_float((unsigned)gb_);
                        if (__all((unsigned)(ga_ >> 32) == want && (unsigned)(gb_ >> 32) == want)) break;
                        if (wave_poll_fail(a, spins, abortf, l, s, t)) break;
                    }
                }
                xpre[128 * wave + lane] = v0; xpre[128 * wave + 64 + lane] = v1;
            }
            __builtin_amdgcn_wave_barrier();
            float mine_pre = 0.f;
            {
                const int lane = lane_now(), cg = lane & 15;
                float acc[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) acc[i] = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xpre + 128 * wave + 64 * m + 4 * cg);
#pragma unroll
                    for (int i = 0; i < 13; ++i) {
                        const int r = i * 8 + 4 * m;
                        acc[i] = fmaf(w[r], xv[0], acc[i]); acc[i] = fmaf(w[r + 1], xv[1], acc[i]); acc[i] = fmaf(w[r + 2], xv[2], acc[i]); acc[i] = fmaf(w[r + 3], xv[3], acc[i]);
                    }
                }
                // lane (g, cg) keeps the past-tap sum of row 13 g + cg for later (one register; cg >= 13: unused)
#pragma unroll
                for (int i = 0; i < 13; ++i) { const float v = row16_sum(acc[i]); mine_pre = (i == 0 || cg == i) ? v : mine_pre; }
                // conditioning columns (modules.py:189-193 conv1x1c): row tid / 8, ten columns per lane, weights from L2 -- nothing waits for this
                float cs = 0.f;
                const int tq = 64 * wave + lane;
                const float* wcrow = a.wcond + ((size_t)(l * NCU + j) * CROWS + (tq >> 3)) * PCIN + (tq & 7) * 10;
                const float* cq = a.cond + ((size_t)s * T + t) * PCIN + (tq & 7) * 10;
#pragma unroll 2
                for (int k = 0; k < 10; ++k) cs = fmaf(wcrow[k], cq[k], cs);                    // (a real loop: twenty loads in flight cost registers the weights do not leave)
                cs += __shfl_xor(cs, 1, 64); cs += __shfl_xor(cs, 2, 64); cs += __shfl_xor(cs, 4, 64);
                if ((tq & 7) == 0) cpart[tq >> 3] = cs;
            }
            // ---- the token arrives in two parts.  x_{l-1}(t) and the skip sum leave stage l - 1 about half a microsecond before z_{l-1}(t) (it publishes them
            // first), so the current tap is done by the time z arrives; behind z: the out / skip rows first (x_l(t) goes out early for the same reason), then
            // the z columns of the gate rows, tanh / sigmoid, z_l(t).  Every wave polls its own granules until all carry the tag -- no block barrier per poll.
            stamp(a, l, j, s, t, 0);
            const u64* p = prev + ((long)s * prl + mp) * TOK;
            int qt = 64 * wave + lane_now();          // (per-thread indices are re-derived where they are used: hoisted out of the loop they cost ~30 registers for its whole length)
            if (l == 0) {
                // the previous sample of this stream.  Teacher-forced steps (t < n_test) take the given input instead but WAIT for the sample all the
                // same: that wait is the pipeline's only back-pressure -- without it stage 0 runs ahead of the later stages and laps its own ring
                float cur = 0.f;
                if (t > 0) {
                    for (unsigned spins = 0;;) {
                        const u64 gq = gload(p);
                        cur = __uint_as_float((unsigned)gq);
                        if ((unsigned)(gq >> 32) == (unsigned)t) break;                            // (one address for the whole wave: uniform)
                        if (wave_poll_fail(a, spins, abortf, l, s, t)) break;
                    }
                }
                if (t < n_test) cur = a.test_inputs[(size_t)s * n_test + t];
                xcur[qt] = fmaf(cur, a.w_first[qt], a.b_first[qt]);            // wavenet.py:118 first_conv
            } else {
                for (unsigned spins = 0;;) {
                    const u64 g0 = gload(p + TOK_X + qt);
                    if (__all((unsigned)(g0 >> 32) == (unsigned)(t + 1))) { xcur[qt] = __uint_as_float((unsigned)g0); break; }
                    if (wave_poll_fail(a, spins, abortf, l, s, t)) break;
                }
            }
            __builtin_amdgcn_wave_barrier();          // (a wave multiplies the 64 columns it fetched itself: no block barrier; an abort is seen at the next one)
            stamp(a, l, j, s, t, 1);
            // ---- current tap: the wave's 64 columns (registers); lane (g, cg) ends up with the sum of row 13 g + cg over past + current taps
            float mine;
            {
                const int cg = lane_now() & 15;
                float acc[13];
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xcur + 64 * wave + 4 * cg);
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    const int r = 104 + 4 * i;
                    acc[i] = fmaf(w[r + 3], xv[3], fmaf(w[r + 2], xv[2], fmaf(w[r + 1], xv[1], w[r] * xv[0])));
                }
#pragma unroll
                for (int i = 0; i < 13; ++i) acc[i] = row16_sum(acc[i]);
                mine = acc[0];
#pragma unroll
                for (int i = 1; i < 13; ++i) mine = cg == i ? acc[i] : mine;
                mine += mine_pre;
            }
            // the register part of row r sits in lane 16 (r / 13) + r % 13; the lane that adds it to the z columns of row r is lane r: moved now, before z arrives

            if (l > 0) {
                // ---- z_{l-1}(t): every wave fetches the 32 values it multiplies (lanes 32 - 63 repeat them)
                for (unsigned spins = 0;;) {
                    const int l31 = lane_now() & 31;
                    const u64 g0 = gload(p + TOK_Z + 32 * wave + l31);
                    if (__all((unsigned)(g0 >> 32) == (unsigned)(t + 1))) { zin[32 * wave + l31] = __uint_as_float((unsigned)g0); break; }
                    if (wave_poll_fail(a, spins, abortf, l, s, t)) break;
                }
                __builtin_amdgcn_wave_barrier();
            }
            stamp(a, l, j, s, t, 2);
            u64* q = ring + ((long)s * rl + m0) * TOK;
            const unsigned tag = (unsigned)(t + 1);
            int ln = lane_now();
            // ---- behind z (LDS rows, lane = row, a wave's 32 columns of z by broadcast reads, two products per v_pk_fma_f32; each loop software-pipelined by hand:
            // the reads of step i + 1 are issued before the products of step i -- left to hipcc, a loop either waits out an LDS round trip per step or, unrolled,
            // hoists all its reads into registers the weights do not leave):
            //   1. the residual rows, all waves; block barrier; WAVE 7 publishes x_l(t) -- half a microsecond ahead of z_l(t), so that the next stage has its
            //      current tap done when z_l(t) lands;
            //   2. meanwhile waves 0 - 6 do the z columns of the gate rows (40 / 32 columns each); block barrier;
            //   3. wave 0 finishes z_l(t).
            auto rowdot = [&](const float* zp, const float* rp, int n8, f32x2 acc) {      // n8 steps of 8 columns, the next step's four reads in flight
                f32x4 zc0 = *reinterpret_cast<const f32x4*>(zp), zc1 = *reinterpret_cast<const f32x4*>(zp + 4);
                f32x4 wc0 = *reinterpret_cast<const f32x4*>(rp), wc1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll 1
                for (int k8 = 1; k8 < n8; ++k8) {
                    const f32x4 zn0 = *reinterpret_cast<const f32x4*>(zp + 8 * k8), zn1 = *reinterpret_cast<const f32x4*>(zp + 8 * k8 + 4);
                    const f32x4 wn0 = *reinterpret_cast<const f32x4*>(rp + 8 * k8), wn1 = *reinterpret_cast<const f32x4*>(rp + 8 * k8 + 4);
                    acc = __builtin_elementwise_fma(wc0.xy, zc0.xy, acc); acc = __builtin_elementwise_fma(wc0.zw, zc0.zw, acc);
                    acc = __builtin_elementwise_fma(wc1.xy, zc1.xy, acc); acc = __builtin_elementwise_fma(wc1.zw, zc1.zw, acc);
                    zc0 = zn0; zc1 = zn1; wc0 = wn0; wc1 = wn1;
                }
                acc = __builtin_elementwise_fma(wc0.xy, zc0.xy, acc); acc = __builtin_elementwise_fma(wc0.zw, zc0.zw, acc);
                acc = __builtin_elementwise_fma(wc1.xy, zc1.xy, acc);
                return __builtin_elementwise_fma(wc1.zw, zc1.zw, acc);
            };
            if (l > 0) {
                const f32x2 b1 = rowdot(zin + 32 * wave, wb + min(ln, XSL - 1) * LROW + 32 * wave, 4, f32x2{0.f, 0.f});   // (lanes past the last row re-read it: a broadcast, nobody reads their sums)
                if (ln < XSL) pb[wave * PBS + ln] = b1.x + b1.y;
            }
            FINE_STAMP(4);
            __syncthreads();                          // (layer 0 too: wave 7 publishes columns of xcur that other waves wrote)
            FINE_STAMP(5);
            if (wave == NW - 1) {
                if (ln < XSL) {
                    const int c = XSL * j + ln;
                    if (c < PC) {
                        float v = xcur[c];
                        if (l > 0) {
                            float sum = bia[56 + ln];
#pragma unroll
                            for (int wv = 0; wv < NW; ++wv) sum += pb[wv * PBS + ln];           // fixed order
                            v = (sum + v) * r5;                                                   // modules.py:204-206
                        }
                        gstore(q + TOK_X + c, tag, v);
                    }
                }
            } else {
                // seven waves share the 256 z columns: waves 0 - 3 take 40 (five steps), waves 4 - 6 take 32 -- one pipelined loop each
                f32x2 ga = {0.f, 0.f};
                if (l > 0) {
                    const int c0 = wave < 4 ? 40 * wave : 160 + 32 * (wave - 4);
                    ga = rowdot(zin + c0, wz + min(ln, GROWS - 1) * LROW + c0, wave < 4 ? 5 : 4, ga);
                }
                if (ln < GROWS) pg[wave * GROWS + ln] = ga.x + ga.y;
            }
            __builtin_amdgcn_wave_barrier();
            if ((ln & 15) < 13) {
                const int r = wave * GROWS + 13 * (ln >> 4) + (ln & 15);
                pg[r] = wave == NW - 1 ? mine : pg[r] + mine;
            }
            FINE_STAMP(6);
            __syncthreads();
            if (*abortf) return;
            FINE_STAMP(7);
            // ---- z_l(t) = tanh(a) sigmoid(b) (modules.py:201): lane r < 26 forms tanh of row r, lane 26 + r the sigmoid of row 26 + r, one shuffle joins them
            qt = 64 * wave + lane_now();
            if (qt < 64) {
                const int r = min(qt, GROWS - 1);
                const float v = ((bia[r] + cpart[r]) + ((pg[r] + pg[GROWS + r]) + (pg[2 * GROWS + r] + pg[3 * GROWS + r]))) +
                                ((pg[4 * GROWS + r] + pg[5 * GROWS + r]) + (pg[6 * GROWS + r] + pg[7 * GROWS + r]));        // fixed order, depth 4
                const bool th = qt < HSL;
                const float e = expf(th ? -2.f * v : -v);                                         // tanh(a) = (1 - e^-2a) / (1 + e^-2a)
                const float f = (th ? 1.f - e : 1.f) / (1.f + e);
                const float sg = __shfl(f, qt + HSL, 64);
                const int h = HSL * j + qt;
                if (th && h < PH) gstore(q + TOK_Z + h, tag, f * sg);
            }
            stamp(a, l, j, s, t, 3);
            // ---- nothing waits for the skip sum before the head: its 26 rows come last (wavenet.py:343-346), the previous stage's sum was fetched with x
            ln = lane_now();
            qt = 64 * wave + ln;
            if (l > 0) {
                f32x2 b2 = {0.f, 0.f};
                const float* rb2 = wb + (XSL + min(ln, SSL - 1)) * LROW + 32 * wave;
#pragma unroll 2
                for (int k4 = 0; k4 < 8; ++k4) {
                    const f32x4 zv = *reinterpret_cast<const f32x4*>(zin + 32 * wave + 4 * k4);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(rb2 + 4 * k4);
                    b2 = __builtin_elementwise_fma(wv.xy, zv.xy, b2); b2 = __builtin_elementwise_fma(wv.zw, zv.zw, b2);
                }
                if (ln < SSL) pb[wave * PBS + 64 + ln] = b2.x + b2.y;
            }
            __syncthreads();                          // (also keeps a fast wave's next token out of cpart / pg / pb / xcur while waves 0 and 7 still publish this one)
            if (qt >= 64 && qt < 128) {                   // wave 1: the skip rows; the previous stage's sum arrives last of all, fetched here by the thread that needs it
                const int k = min(qt - 64, SSL - 1), si = min(SSL * j + k, PS - 1);
                float v = 0.f;
                if (l > 0) {
                    float sum = bia[56 + XSL + k];
#pragma unroll
                    for (int wv = 0; wv < NW; ++wv) sum += pb[wv * PBS + 64 + k];
                    float prevsum = 0.f;
                    if (l > 1) {
                        for (unsigned spins = 0;;) {
                            const u64 g1 = gload(p + TOK_S + si);
                            prevsum = __uint_as_float((unsigned)g1);
                            if (__all((unsigned)(g1 >> 32) == (unsigned)(t + 1))) break;
                            if (wave_poll_fail(a, spins, abortf, l, s, t)) break;
                        }
                    }
                    v = l == 1 ? sum : (prevsum + sum) * r5;
                }
                if (qt - 64 < SSL && SSL * j + (qt - 64) < PS) gstore(q + TOK_S + si, tag, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ head stages
// rows of a 256-wide 1x1 layer held in LDS; `mode` 0: v = relu((skin + W z + b) sqrt(.5)) (the last layer's skip rows), 1: v = relu(W x + b)
__device__ __forceinline__ void dense_stage(const WnPipe& a, const int st, const int j, const int mode, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wr = lds;                  // [64][256]
    float* xin = wr + 64 * 256;       // [256]
    float* skin = xin + 256;          // [64]
    float* bia = skin + 64;           // [64]
    int* abortf = reinterpret_cast<int*>(bia + 64);
    if (tid == 0) *abortf = 0;
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
            stamp(a, st, j, s, t, 0);
            if (wave < 4) {                           // every wave polls its own 64 granules: no block barrier per poll
                for (unsigned spins = 0;;) {
                    const u64 g = gload(p + in_off + tid);
                    if (__all((unsigned)(g >> 32) == (unsigned)(t + 1))) { xin[tid] = __uint_as_float((unsigned)g); break; }
                    if (wave_poll_fail(a, spins, abortf, st, s, t)) break;
                }
            }
            __syncthreads();
            if (*abortf) return;
            stamp(a, st, j, s, t, 1);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xin + 4 * lane);
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = dot4(*reinterpret_cast<const f32x4*>(wr + (wave * 8 + i) * 256 + 4 * lane), xv, 0.f);
            u64* q = slot_of(a, st, s, t);
            float mine = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float v = wave_sum_dpp(acc[i]); mine = lane == i ? v : mine; }
            if (lane < 8) {
                const int k = wave * 8 + lane;
                float y = mine + bia[k];
                if (mode == 0) {
                    // the skip sum of stage 23 is the last thing it publishes: fetched here, behind the products, by the lane that adds it
                    float prevsum = 0.f;
                    for (unsigned sp2 = 0;;) {
                        const u64 g1 = gload(p + TOK_S + 64 * j + k);
                        prevsum = __uint_as_float((unsigned)g1);
                        if ((unsigned)(g1 >> 32) == (unsigned)(t + 1)) break;
                        if (++sp2 > SPIN_LIMIT) { if (atomicCAS(a.err, 0u, 1u) == 0u) { a.err[1] = (unsigned)st; a.err[2] = (unsigned)s; a.err[3] = (unsigned)t; } break; }
                        if ((sp2 & 63u) == 0u && eload(a.err) != 0u) break;
                    }
                    y = (prevsum + y) * r5;                                                       // wavenet.py:343-346, then :349 ReLU
                }
                gstore(q + 64 * j + k, (unsigned)(t + 1), y > 0.f ? y : 0.f);
            }
            stamp(a, st, j, s, t, 3);
            __syncthreads();                          // xin / skin are rewritten by the next poll
        }
}

// the head's second 1x1 (30 rows) and the sampler (mixture.py:117-153 with injected uniforms)
__device__ __forceinline__ void sample_stage(const WnPipe& a, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wr = lds;                  // [32][256]
    float* xin = wr + 32 * 256;       // [256]
    float* yo = xin + 256;            // [32]
    float* bia = yo + 32;             // [32]
    float* us = bia + 32;             // [16] the sampler's noise terms
    int* abortf = reinterpret_cast<int*>(us + 16);
    if (tid == 0) *abortf = 0;
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
            stamp(a, NL + 2, 0, s, t, 0);
            // the sampler's noise does not depend on the logits: formed while the token is on its way (same expressions as mixture.py:125-150, so the same bits)
            if (tid < K3) us[tid] = logf(-logf(a.u1[((size_t)s * a.T + t) * K3 + tid]));
            if (tid == 32) { const float u = a.u2[(size_t)s * a.T + t]; us[K3] = logf(u) - logf(1.f - u); }
            if (wave < 4) {
                for (unsigned spins = 0;;) {
                    const u64 g = gload(p + tid);
                    if (__all((unsigned)(g >> 32) == (unsigned)(t + 1))) { xin[tid] = __uint_as_float((unsigned)g); break; }
                    if (wave_poll_fail(a, spins, abortf, NL + 2, s, t)) break;
                }
            }
            __syncthreads();
            if (*abortf) return;
            stamp(a, NL + 2, 0, s, t, 1);
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
                    const float v = yo[k] - us[k];
                    if (v > best) { best = v; arg = k; }
                }
                const float m = yo[K3 + arg], ls = fmaxf(yo[2 * K3 + arg], a.log_scale_min);
                float x = m + expf(ls) * us[K3];
                x = fminf(fmaxf(x, -1.f), 1.f);
                a.out[(size_t)s * a.T + t] = x;
                gstore(slot_of(a, NL + 2, s, t), (unsigned)(t + 1), x);
                stamp(a, NL + 2, 0, s, t, 3);
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

constexpr size_t PIPE_LDS_BYTES = (size_t)((GROWS + BROWS) * LROW + KPRE + PC + PH + 32 + 64 + 136 + PCIN + NW * GROWS + NW * PBS + 4) * sizeof(float);

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------------------
extern "C" int viai_wn_pipe_ok(const viai_wn_synth* s) {
    if (!s || s->C != PC || s->G != 2 * PH || s->S != PS || s->cin != PCIN || s->n_layers != NL || s->out_ch != 30 || s->out_ch % 3 != 0) return 0;
    if (s->B < 1 || s->B > PIPE_MAXB || s->cond == nullptr) return 0;
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
    case 1: return (long)NL * NCU * (GROWS + BROWS) * LROW;      // wlds
    case 2: return (long)NL * NCU * 136;                         // bias
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

static u64* g_prof = nullptr;
static int g_prof_t = -1;
// debug aid: the next viai_wn_pipe_run calls stamp time step t (100 MHz wall clock) into buf, [27 stages][8 streams][4] uint64; buf = NULL switches it off
extern "C" int viai_wn_pipe_profile(void* buf, int t) { g_prof = (u64*)buf; g_prof_t = t; return 0; }

// time steps [t0, t0 + n) of every stream in ONE launch.  `tok` must be zero before the call with t0 == 0 and carried over between calls;
// err: 4 zeroed uint32 (err[0] != 0 after the call: 1 = a wait timed out at (stage, stream, t) = err[1..3]; 2 is no longer raised: a past tap is waited for like a token).
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
    a.prof = g_prof; a.prof_t = g_prof_t;
    a.B = s->B; a.T = s->T; a.n_test = s->n_test; a.t0 = t0; a.t1 = t0 + n_steps; a.out_ch = s->out_ch; a.log_scale_min = s->log_scale_min;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wn_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIPE_LDS_BYTES) != hipSuccess) return (int)hipGetLastError();
        attr_set = true;
    }
    VIAI_LAUNCH(wn_pipe_kernel, dim3(256), dim3(NT), PIPE_LDS_BYTES, (hipStream_t)stream, a);
    return viai_launch_status();
}
