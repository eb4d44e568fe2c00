// Patch-staged convolution kernels whose activation operand arrives by LDS-DMA (`buffer_load_dwordx4 ... lds`), gfx950, round 5.
//
// A P16 tensor (viai_bf3.h) already holds, per pixel and 32-channel group, the two fp16 planes the f16x2 kernels multiply: 128 bytes =
// eight 16-byte pieces (pieces 0..3 = leading terms of channels 8 k .. 8 k + 7, pieces 4..7 = their remainders), and a piece is exactly
// what one lane of `v_mfma_f32_32x32x16_f16` takes as its A operand.  So nothing has to pass through registers on the way to LDS: a
// wave-instruction copies eight pixels (8 x 128 bytes = 1 KiB, lane = pixel * 8 + slot) straight into the stage, asynchronously, and
// the only per-element work left in the kernel is the MFMA itself.  What that buys over the register-staged kernels
// (conv_halo_bf3.hip): (i) no staging registers, so the ring can be THREE tiles deep -- the register-staged kernels had one tile of
// loads in flight per block and were bound by memory latency, not bandwidth (G.convblock5: 67 MB in 25 us = 2.7 TB/s); (ii) no
// ds_write pass and no per-quad VALU.
//
// LDS image of a stage: [patch pixel][128 bytes], the pixel's eight pieces permuted: piece q sits in slot q ^ ((column >> 1) & 7).
// A `ds_read_b128` is serviced in groups of 16 lanes that together must touch all 64 banks once (MI355X_MICROARCH.md, LDS): the 16
// lanes of a group read 16 consecutive patch columns (the MFMA row -> pixel map below), same piece q; a pixel record is half a bank
// row, so eight lanes share each half and the XOR with (column >> 1) & 7 spreads them over its eight slots.  LDS-DMA writes lane-linear,
// so the permutation is applied on the SOURCE side: lane (pixel, slot) fetches piece slot ^ f(column) of that pixel (same 128-byte
// line, so the global access stays fully coalesced).  Pixels outside the image are lanes with an out-of-range offset: the DMA
// writes zeros for them (tools/probes/glds_oob.hip, gpurun_out/r04_b/glds_oob.txt), which is the convolution's zero padding.
//
// The DMA is issued from inline asm (M0 = LDS destination, saved and restored around the batch): hipcc's waitcnt pass does not see
// it, which is the point -- a compiler-visible LDS-DMA makes every later barrier drain `vmcnt(0)` (cdna_hip_programming.md,
// "Pipelining across barriers").  Completion is counted by hand: after the MFMAs of tile k the wave waits `vmcnt(<DMA instructions of
// the batch it issued at the top of this iteration>)`, i.e. until everything older than that batch -- tile k + 1's data -- has landed,
// BEFORE it issues its epilogue stores (which share the counter); the barrier at the top of iteration k + 1 then covers the other waves'
// parts.  Compiler-counted waits (the filter fetch of the prologue, nothing in the loop) can only over-wait beside invisible younger
// operations, never under-wait.
//
// Reference call sites: the 32 -> 32 channel ConvTranspose2d layers of the decoder (networks/New_Inpainting_Networks.py:61-63,85-88:
// convblock4_1/2, convblock5_0..3, conv6_1) and their autograd data gradients.
#include "viai_common.h"
#include "viai_internal.h"
#include "viai_bf3.h"
#include <cstdlib>

// persistent grid of the loader / consumer kernels: 256 blocks = one per compute unit.  A block of these kernels takes a CU's whole register file, so
// while one launch is resident NOTHING of another stream starts anywhere on the chip until a block exits -- in the vision-infused step, where two ResNet
// chains share the chip, the other chain's 10 us BatchNorm-finalize launches then wait hundreds of microseconds (profiles/r05_f_av_kernel_stats.csv:
// 238 us as run, 10 us alone).  VIAI_DMA_GRID (tuning switch) leaves some compute units to the other streams.
static int viai_dma_grid() {
    static int g = -1;
    if (g < 0) { const char* e = getenv("VIAI_DMA_GRID"); g = e ? atoi(e) : 256; if (g < 16 || g > 256) g = 256; }
    return g;
}

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int DT_H = 8, DT_W = 16;                      // output tile: 128 pixels = 4 waves x 32 MFMA rows
constexpr int DP_H = DT_H + 2, DP_W = DT_W + 2;         // staged patch: 10 x 18 pixels
constexpr int DP_NPIX = DP_H * DP_W;                    // 180
constexpr int DP_INSTR = 24;                            // DMA wave-instructions per stage (8 pixels each): 192 pixel slots, 6 per wave
constexpr int DP_STAGE = DP_INSTR * 1024;               // 24 KB
constexpr int DP_NSTG = 3;
constexpr int DP_OOB = 0x7fffffff;

// wave-uniform raw buffer descriptor in SGPRs (same words as __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000))
__device__ __forceinline__ i32x4 rsrc_sgpr(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull));
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffull));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// q = n / d for n * d < 2^32 with magic = ceil(2^32 / d) (host side: tile_magic): one s_mul_hi_u32 instead of a ~25-instruction division
__device__ __forceinline__ int div_magic(int n, unsigned magic) { return magic == 0u ? n : (int)__umulhi((unsigned)n, magic); }      // (magic 0: d = 1)
struct TileDec {
    int tiles_x, tiles_y; unsigned mx, my;                    // tiles per map row / column and their magics
#ifdef VIAI_PROF
    unsigned long long* prof;                                  // tools/probes/halo_dma_bench.hip: [block][16 tiles][8 stamps] shader-clock stamps of wave 0
#endif
};
#ifdef VIAI_PROF
#define PROF_STAMP(k, i) do { if (tid == 0 && (k) < 16) td.prof[((size_t)blockIdx.x * 16 + (k)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PROF_STAMP(k, i) do { } while (0)
#endif

// six LDS-DMA wave-instructions: lane l of instruction j copies the 16 bytes at (buffer + soff + v_j[l]) to LDS byte lds0 + j * 4096 + 16 l
// (instruction i = wave + 4 j of a stage, 1 KiB each).  `s_nop 4`: the descriptor / offsets may come straight from v_readfirstlane;
// `s_nop 0` after every M0 write (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void dma_batch6(const i32x4& rs, unsigned lds0, int soff, int v0, int v1, int v2, int v3, int v4, int v5) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %4, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %5, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %6, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %7, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %8, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %9, %1, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(rs), "s"(lds0), "s"(soff), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5)
        : "memory", "scc");
}

struct DmaSlots { int s[9]; };                         // weight slot of window position (row * 3 + col)

// 32 -> (<= 32) channel 3 x 3 stride-1 layer on a P16 input; the whole filter in registers (as conv_halo_f16_c32_kernel), the input
// patch of tile k + 2 in flight while tile k is multiplied.  Same MFMA order and epilogue as the register-staged kernel: bit-identical.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_halo_c32_dma_kernel(const ConvArgs a, int hy0, int hx0, int ntiles, DmaSlots slots, TileDec td) {
    constexpr int KS = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_d[];      // [3 stages][24 KB] + [4 waves][2][32] floats
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float ascale = f16_scale_from_amax(a.amax);
    auto decode = [&](int tile, int& tx, int& ty, int& n) {      // tile -> (image, tile row, tile column)
        const int r = div_magic(tile, td.mx);
        tx = tile - r * td.tiles_x;
        n = div_magic(r, td.my);
        ty = r - n * td.tiles_y;
    };

    const long in_bytes = (long)g.N * g.IH * g.IW * 128;
    const i32x4 rs_in = rsrc_sgpr(a.in, (unsigned)in_bytes);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_d;

    // DMA lane map: instruction i = wave + 4 j covers patch pixels 8 i .. 8 i + 7; lane -> (pixel 8 i + (lane >> 3), slot lane & 7)
    int vint[6], prc[6];                                // interior-tile offset (relative to the patch origin) / (row << 8) | col, -1 beyond the patch
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int p = 8 * (wave + 4 * j) + (lane >> 3);
        const int pr_ = p / DP_W, pc_ = p - pr_ * DP_W;
        const int q = (lane & 7) ^ ((pc_ >> 1) & 7);
        prc[j] = p < DP_NPIX ? (pr_ << 8) | pc_ : -1;
        vint[j] = p < DP_NPIX ? (pr_ * g.IW + pc_) * 128 + q * 16 : DP_OOB;
    }
    auto issue = [&](int tile, int stage) {             // tile, stage wave-uniform
        int tx, ty, n;
        decode(tile, tx, ty, n);
        const int py = ty * DT_H + hy0, px = tx * DT_W + hx0;
        const unsigned lds0 = lds_base + stage * DP_STAGE + wave * 1024;
        const bool interior = py >= 0 && px >= 0 && py + DP_H <= g.IH && px + DP_W <= g.IW;
        if (interior) {
            const int soff = __builtin_amdgcn_readfirstlane(((n * g.IH + py) * g.IW + px) * 128);
            dma_batch6(rs_in, lds0, soff, vint[0], vint[1], vint[2], vint[3], vint[4], vint[5]);
        } else {
            int v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int iy = py + (prc[j] >> 8), ix = px + (prc[j] & 255);
                const int dm = ((g.IH - 1 - iy) | iy | (g.IW - 1 - ix) | ix | prc[j]) >> 31;          // all ones: outside the image / beyond the patch
                const int q = (lane & 7) ^ (((prc[j] & 255) >> 1) & 7);
                v[j] = ((((n * g.IH + iy) * g.IW + ix) * 128 + q * 16) & ~dm) | (dm & DP_OOB);
            }
            dma_batch6(rs_in, lds0, 0, v[0], v[1], v[2], v[3], v[4], v[5]);
        }
    };

    const int G = gridDim.x;
    const int t0 = blockIdx.x;
    // the first tile on its way before anything else (the second follows the filter: see below)
    if (t0 < ntiles) issue(xcd_remap(t0, ntiles), 0);

    // the whole filter, once per block: wf[position][k-step][plane]
    const int frag_plane = g.wtaps * KS * 1024;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 2 * frag_plane, 0x00020000);
    u32x4 wf[9][KS][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                wf[t][ks][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, (slots.s[t] * KS + ks) * 1024 + p * frag_plane, 0);

    // MFMA row i = lane & 31 -> tile pixel (2 * wave + (i >> 4), i & 15); lane >> 5 = k half.  Fragment address of window column tx, k-step ks
    const int pr = 2 * wave + ((lane & 31) >> 4), pc = lane & 15, kh = lane >> 5;
    int aoff[3][KS];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            aoff[tx][ks] = (pr * DP_W + pc + tx) * 128 + (((2 * ks + kh) ^ (((pc + tx) >> 1) & 7)) * 16);
    const int half = lane >> 5, col = lane & 31;
    const float bv = a.bias != nullptr ? a.bias[col] : 0.f;
    const float inv = 1.0f / (ascale * F16_WSCALE);
    float* red = reinterpret_cast<float*>(smem_d + DP_NSTG * DP_STAGE);          // [4 waves][2][32]
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)((long)g.N * g.OH * g.OW * 128), 0x00020000);
    const int orow = g.OW * 128;                                                  // bytes per output row (32 channels)
    const int ovoff = 2 * wave * orow + (4 * half) * 128 + col * 4;

    // The filter fetch is the one compiler-counted load group of this kernel, and its last `vmcnt(0)` would sit in front of the first tile's
    // last MFMAs -- draining every DMA issued before it.  So the values are pinned here (the compiler waits for them HERE, together with
    // tile 0, which is older), and only then does tile 1 go out: compute starts after one tile's worth of the start-up burst instead of two.
#pragma unroll
    for (int t = 0; t < 9; ++t)
        asm volatile("" : "+v"(wf[t][0][0]), "+v"(wf[t][0][1]), "+v"(wf[t][1][0]), "+v"(wf[t][1][1]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t0 + G < ntiles) issue(xcd_remap(t0 + G, ntiles), 1);
    int stage = 0;
    [[maybe_unused]] int kk = 0;
    PROF_STAMP(0, 7);
    for (int it = t0; it < ntiles; it += G) {
        const int tile = xcd_remap(it, ntiles);
        PROF_STAMP(kk, 0);
        __syncthreads();                               // every wave's part of this tile has landed; stage + 2 (tile k - 1) is free
        PROF_STAMP(kk, 1);
        const bool more2 = it + 2 * G < ntiles;
        if (more2) issue(xcd_remap(it + 2 * G, ntiles), stage >= 1 ? stage - 1 : 2);
        const unsigned char* Sb = smem_d + stage * DP_STAGE;
        PROF_STAMP(kk, 2);

        f32x16 c0, c1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned char* As = Sb + aoff[t % 3][ks] + (t / 3) * (DP_W * 128);
                const f16x8 a1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(As));
                const f16x8 a2 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(Sb + (aoff[t % 3][ks] ^ 64) + (t / 3) * (DP_W * 128)));
                const f16x8 b1 = __builtin_bit_cast(f16x8, wf[t][ks][0]), b2 = __builtin_bit_cast(f16x8, wf[t][ks][1]);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, c0, 0, 0, 0);      // small terms on one chain
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, c0, 0, 0, 0);
            }
        f32x16 acc = (c0 + c1) * inv;

        PROF_STAMP(kk, 3);
        // tile k + 1 (issued one iteration ago) must have landed before the next barrier: everything older than this iteration's batch
        if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        PROF_STAMP(kk, 4);
        // epilogue: the lane's sixteen values go to pixels (2 wave + (e >> 3), (e & 3) + 8 ((e >> 2) & 1) + 4 half) of the tile, channel col:
        // one buffer store each, the per-lane part of the address in voff, the rest wave-uniform (scalar base + immediate)
        int tx, ty, n;
        decode(tile, tx, ty, n);
        const int obase = __builtin_amdgcn_readfirstlane((((n * g.OH + ty * DT_H) * g.OW + tx * DT_W) * 32) * 4);
        auto store16 = [&](auto ACT) {                  // (the activation decision once per tile, not once per element)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[e] + bv;
                if constexpr (decltype(ACT)::value) v = viai_act(v, a.act, a.slope);
                acc[e] = v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, ovoff + ((e & 3) + 8 * ((e >> 2) & 1)) * 128,
                                                      (e >> 3) ? obase + orow : obase, 0);
            }
        };
        if (a.stat == nullptr && a.act != VIAI_ACT_NONE) store16(std::true_type{});
        else store16(std::false_type{});
        PROF_STAMP(kk, 5);
        if (a.stat != nullptr) {          // block-local (mean, M2) over the 128 pixels of this tile (as conv_halo_f16_c32_kernel)
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[e];
            t += __shfl_xor(t, 32, 64);
            const float mw = t * (1.f / 32.f);
            float m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = acc[e] - mw; m2 += d * d; }
            m2 += __shfl_xor(m2, 32, 64);
            if (half == 0) { red[(wave * 2 + 0) * 32 + col] = mw; red[(wave * 2 + 1) * 32 + col] = m2; }
            __syncthreads();
            if (wave == 0 && half == 0) {
                const float m0 = red[0 * 32 + col], m1 = red[2 * 32 + col], m2_ = red[4 * 32 + col], m3 = red[6 * 32 + col];
                const float mean = 0.25f * ((m0 + m1) + (m2_ + m3));
                float M2 = (red[1 * 32 + col] + red[3 * 32 + col]) + (red[5 * 32 + col] + red[7 * 32 + col]);
                M2 += 32.f * (((m0 - mean) * (m0 - mean) + (m1 - mean) * (m1 - mean)) + ((m2_ - mean) * (m2_ - mean) + (m3 - mean) * (m3 - mean)));
                a.stat[(size_t)col * a.nblk_m + tile] = mean;
                a.stat[(size_t)(a.Cout + col) * a.nblk_m + tile] = M2;
            }
            // (red[] is rewritten only after the next iteration's top barrier)
        }
        PROF_STAMP(kk, 6);
        ++kk;
        stage = stage == 2 ? 0 : stage + 1;
    }
}

}  // namespace

#ifdef VIAI_PROF
static unsigned long long* viai_dma_prof_buf = nullptr;
#endif
static unsigned tile_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

// P16 input, 32 -> 32 channels into one destination, byte offsets and tile counts within the 32-bit arithmetic of the kernel
bool viai_conv_halo_c32_dma_ok(const ConvArgs& a) {
    if (!viai_halo_dma_on() || !a.in_p16 || a.amax == nullptr || a.C1 != 32 || a.C2 != 0 || a.Cout != 32 || a.OC1 != 32) return false;
    const ConvGeom& g = a.g;
    if ((long)g.N * g.IH * g.IW * 128 >= (1l << 31) || (long)g.N * g.OH * g.OW * 128 >= (1l << 31)) return false;
    const long tiles = (long)g.N * (g.OH / DT_H) * (g.OW / DT_W);
    return g.OH % DT_H == 0 && g.OW % DT_W == 0 && tiles * 64 < (1l << 31);
}

// P16 launches of the register-filter halo kernel's layers (viai_conv_halo16_ok): called by viai_conv_halo_bf3_launch
int viai_conv_halo_c32_dma_launch(ConvArgs& a, int y0, int x0, const int* slots9, hipStream_t st) {
    if (!viai_conv_halo_c32_dma_ok(a)) return (int)hipErrorInvalidValue;
    constexpr int lds = DP_NSTG * DP_STAGE + 4 * 2 * 32 * (int)sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_c32_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    DmaSlots sl;
    for (int t = 0; t < 9; ++t) sl.s[t] = slots9[t];
    a.nblk_m = a.M / 128;
    a.nblk_n = 1;
    int grid = 256 * 2;
    if (grid > a.nblk_m) grid = a.nblk_m;
    TileDec td;
    td.tiles_x = a.g.OW / DT_W; td.tiles_y = a.g.OH / DT_H;
    td.mx = tile_magic(td.tiles_x); td.my = tile_magic(td.tiles_y);
#ifdef VIAI_PROF
    td.prof = viai_dma_prof_buf;
#endif
    VIAI_LAUNCH(conv_halo_c32_dma_kernel, dim3(grid), dim3(256), lds, st, a, y0, x0, a.nblk_m, sl, td);
    return viai_launch_status();
}

// bring-up switch (A/B against the register-staged kernels): VIAI_HALO_DMA=0
bool viai_halo_dma_on() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("VIAI_HALO_DMA"); on = e ? atoi(e) : 1; }
    return on != 0;
}

// =====================================================================================================================================
// 3 x 3 STRIDE-2 forward conv on a P16 input, producer / consumer waves (round 5).
//
// Why not the plain LDS-DMA form of the kernel above: here the filter does not fit in registers, so every consumer wave streams weight
// fragments from L2 all through the K loop -- and `vmcnt` retires in ORDER: a weight fragment requested after an activation DMA cannot be
// consumed before that DMA has landed (HBM latency).  The register-staged kernel (conv_halo_wide_f16_kernel<.., 2>) has the same coupling
// between its activation rows and its fragment stream.  The counters are per WAVE, so the two streams go to different waves:
//
//   * 4 LOADER waves (one per SIMD) issue nothing but LDS-DMA: the input patch of one 16-channel k-step -- 17 x 33 pixels x 64 bytes (the
//     k-step's two leading and two remainder pieces of each pixel) -- into a three-deep ring, two stages ahead of the consumers;
//   * 8 CONSUMER waves (2 x 4: 64 pixels x 32 channels each, two per SIMD) read A fragments from the ring (`ds_read_b128`), stream their
//     own weight fragments (compiler-counted loads: nothing else is in their queue) and run 54 MFMAs per stage;
//   * one `s_barrier` per stage joins them: behind it stage q has landed (the loaders waited for it) and the consumers are done with
//     stage q - 1, whose slot the loaders refill with stage q + 2.
// The ring runs across work items (8 x 16 output pixels x 128 channels each; a persistent block walks its items), so the loaders stream
// the next item's patch under this item's epilogue.
//
// Stage image: the patch as four parity sub-patches P[py][px][r][c] = patch(2 r + py, 2 c + px) (window position (ty, tx) of output
// pixel (r, c) is P[ty & 1][tx & 1] at (r + (ty >> 1), c + (tx >> 1)): consecutive output pixels read consecutive records), rows padded to
// 20 / 16 records (multiples of the 256-byte bank row), record = 64 bytes = four 16-byte pieces {lead k-half 0, lead k-half 1, rem 0,
// rem 1} stored at position piece ^ ((c >> 2) & 3): the 16 lanes of a `ds_read_b128` group read 16 consecutive columns, four per bank-row
// quarter, and the XOR gives those four distinct positions -- conflict-free, checked with SQ_LDS_BANK_CONFLICT.
// Reference call sites: networks/Discriminator_Networks.py:20-27 (conv2_1, conv2_2), networks/Inpainting_Networks.py:55-63 (conv3, conv4).
namespace {

constexpr int S2_TH = 8, S2_TW = 16;                                   // output tile
constexpr int S2_P0 = 0, S2_P1 = 9 * 20, S2_P2 = S2_P1 + 9 * 16, S2_P3 = S2_P2 + 8 * 20;    // stride 2: record index of each parity sub-patch
constexpr int S2_NSTG = 3;
constexpr int S2_NLOAD = 4, S2_NCONS = 4;                              // waves: one loader and one consumer per SIMD
constexpr int S2_TM = 4;                                               // 32-pixel MFMA tiles per consumer wave: all 128 pixels of the tile
constexpr int S2_THREADS = 64 * (S2_NLOAD + S2_NCONS);
// S = 2: 17 x 33 patch as four parity sub-patches (612 records, 40 DMA instructions per stage).  S = 1 (round 5, the stride-1 wide layers: D.conv3 forward
// and data gradient): 10 x 18 patch, rows of 20 records (200 records, 16 instructions).  TN = 32-channel tiles per consumer wave: the block covers
// 128 TN output channels.
template <int S, int TN>
struct WideDma {
    static constexpr int NREC = S == 2 ? S2_P3 + 8 * 16 : 10 * 20;
    static constexpr int NPL = S == 2 ? 10 : 4;                        // DMA instructions per loader wave and stage (16 records each)
    static constexpr int STAGE = 4 * NPL * 1024;
    static constexpr int CW = 128 * TN;                                // channels of the block
    static constexpr int OUT = S2_NSTG * STAGE, OUT_BYTES = 64 * CW * 4;   // behind the ring: half an output tile (64 pixels x CW channels fp32) ...
    static constexpr int STAT = OUT + OUT_BYTES;                       // ... and the BatchNorm partials of the tile: [block][mean | M2][CW]
    static constexpr int BIAS = STAT + 2 * 2 * CW * 4;                 // ... and the bias of the block's channels (the loaders apply it: see the epilogue)
    static constexpr int LDS = BIAS + CW * 4;
    static constexpr int NOB = 16 * TN;                                // 1 KB output pieces per loader wave and item
};

struct S2Args {
    int tiles_x, tiles_y; unsigned mx, my;                            // 8 x 16 tiles per map and their division magics
    int nnb; unsigned mnb;                                             // 128-channel blocks per tile (item = tile * nnb + nb)
    int nitems;
    int slot[9];                                                       // weight slot of window position (row * 3 + col)
#ifdef VIAI_PROF
    unsigned long long* prof;                                          // [block][2 roles][32 stages][4 stamps]
#endif
};
#ifdef VIAI_PROF
#define S2_STAMP(role, q, i) do { if (lane == 0 && (wave == 0 || wave == S2_NCONS) && (q) < 32) sa.prof[(((size_t)blockIdx.x * 2 + (role)) * 32 + (q)) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S2_STAMP(role, q, i) do { } while (0)
#endif

// ten LDS-DMA wave-instructions of one loader wave: instruction j -> LDS lds0 + j * 4096 (instruction index = loader + 4 j)
__device__ __forceinline__ void dma_batch10(const i32x4& rs, unsigned lds0, int soff, const int (&v)[10]) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %4, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %5, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %6, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %7, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %8, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %9, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %10, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %11, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %12, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %13, %1, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(rs), "s"(lds0), "s"(soff), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9])
        : "memory", "scc");
}

__device__ __forceinline__ void dma_batch4(const i32x4& rs, unsigned lds0, int soff, const int (&v)[4]) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %4, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %5, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %6, %1, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %7, %1, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(rs), "s"(lds0), "s"(soff), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])
        : "memory", "scc");
}
__device__ __forceinline__ void dma_batch(const i32x4& rs, unsigned lds0, int soff, const int (&v)[10]) { dma_batch10(rs, lds0, soff, v); }
__device__ __forceinline__ void dma_batch(const i32x4& rs, unsigned lds0, int soff, const int (&v)[4]) { dma_batch4(rs, lds0, soff, v); }

template <int S, int TN>
__global__ __launch_bounds__(S2_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wide_dma_kernel(const ConvArgs a, const S2Args sa) {
    using W = WideDma<S, TN>;
    constexpr int NPL = W::NPL, CW = W::CW, NOB = W::NOB;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_s2[];     // [3 stages][W::STAGE] [half output tile] [partials]
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.C1, k16 = Cin / 16;                                        // k-steps per item
    const int G = gridDim.x;
    const int nmine = (sa.nitems - (int)blockIdx.x + G - 1) / G;                  // work items of this block: blockIdx.x + k G
    const int nstage = nmine * k16;
    auto item_of = [&](int k, int& tx, int& ty, int& n, int& nb) {
        const int item = xcd_remap(blockIdx.x + k * G, sa.nitems);
        const int tile = div_magic(item, sa.mnb);
        nb = item - tile * sa.nnb;
        const int r = div_magic(tile, sa.mx);
        tx = tile - r * sa.tiles_x;
        n = div_magic(r, sa.my);
        ty = r - n * sa.tiles_y;
    };

#ifdef VIAI_PROF
    if (tid == 0) sa.prof[(((size_t)blockIdx.x * 2 + 1) * 32 + 31) * 4 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (wave >= S2_NCONS) {
        // ------------------------------------------------------------------------------------------------ loader waves
        const int lw = wave - S2_NCONS;
#ifndef VIAI_S2_LOADER_PRIO
#define VIAI_S2_LOADER_PRIO 0
#endif
        __builtin_amdgcn_s_setprio(VIAI_S2_LOADER_PRIO);   // (raised priority was measured: the loaders then steal issue slots from the one consumer wave of their SIMD, 64.4 -> 69.9 us)
        const i32x4 rs_in = rsrc_sgpr(a.in, (unsigned)((long)g.N * g.IH * g.IW * Cin * 4));
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_s2 + lw * 1024;
        // lane -> (record 16 i + (lane >> 2), position lane & 3) of instruction i = lw + 4 j
        int vint[NPL], edge[NPL];                    // interior offset relative to the patch origin; bits 0 .. 3: patch row 0, column 0, last row, last column; -1: no pixel
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int L = 16 * (lw + 4 * j) + (lane >> 2);
            int prow, pcol, c; bool pix;
            if constexpr (S == 2) {
                int par, rec;
                if (L < S2_P1) { par = 0; rec = L; } else if (L < S2_P2) { par = 1; rec = L - S2_P1; } else if (L < S2_P3) { par = 2; rec = L - S2_P2; } else { par = 3; rec = L - S2_P3; }
                const int py = par >> 1, px = par & 1, pitch = px ? 16 : 20;
                const int r = rec / pitch;
                c = rec - r * pitch;
                pix = L < W::NREC && c < (px ? 16 : 17);
                prow = 2 * r + py; pcol = 2 * c + px;
            } else {
                prow = L / 20; c = L - prow * 20; pcol = c;
                pix = L < W::NREC && c < 18;
            }
            const int qp = (lane & 3) ^ ((c >> 2) & 3);                          // piece held at this position: plane qp >> 1, k-half qp & 1
            vint[j] = pix ? (prow * g.IW + pcol) * Cin * 4 + (qp >> 1) * 64 + (qp & 1) * 16 : DP_OOB;
            edge[j] = pix ? (prow == 0 ? 1 : 0) | (pcol == 0 ? 2 : 0) | ((S == 1 && prow == 9) ? 4 : 0) | ((S == 1 && pcol == 17) ? 8 : 0) : -1;
        }
        auto issue = [&](int q) {                    // stage q = (item q / k16, k-step q % k16) -> ring slot q % 3
            const int k = q / k16, kk = q - k * k16;
            int tx, ty, n, nb;
            item_of(k, tx, ty, n, nb);
            const int py0 = S * ty * S2_TH - 1, px0 = S * tx * S2_TW - 1;       // patch origin (pad 1)
            const unsigned lds0 = lds_base + (q % S2_NSTG) * W::STAGE;
            const int koff = (kk >> 1) * 128 + (kk & 1) * 32;                    // chunk, k-step within the chunk's 128-byte record
            // tile at the map's edge: the pad rows / columns of the patch (stride 2 with the input exactly twice the output: top and left only)
            const int m = (ty == 0 ? 1 : 0) | (tx == 0 ? 2 : 0) | ((S == 1 && ty == sa.tiles_y - 1) ? 4 : 0) | ((S == 1 && tx == sa.tiles_x - 1) ? 8 : 0);
            if (m == 0) {
                const int soff = __builtin_amdgcn_readfirstlane(((n * g.IH + py0) * g.IW + px0) * Cin * 4 + koff);
                dma_batch(rs_in, lds0, soff, vint);
            } else {                                  // those lanes go out of range (zeros)
                const int base = ((n * g.IH + py0) * g.IW + px0) * Cin * 4;
                int v[NPL];
#pragma unroll
                for (int j = 0; j < NPL; ++j) v[j] = (edge[j] < 0 || (edge[j] & m)) ? DP_OOB : vint[j] + base;
                dma_batch(rs_in, lds0, koff, v);
            }
        };
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)((long)g.N * g.OH * g.OW * a.Cout * 4), 0x00020000);
        int kl = 0;                                  // k-step of stage q
        u32x4 obuf[NOB];                             // the last finished item's tile part of this wave: 2 halves x NOB / 2 pieces (a row of 16 pixels x CW channels each half)
        int obase[2] = {0, 0}, pend = NOB;           // their addresses; next piece to store (NOB: none pending)
        // piece P = 64 j + lane of a row (16 pixels x CW / 4 pieces): pixel P / (CW / 4), 16-byte piece P % (CW / 4)
        const int ovoff = TN == 1 ? (lane >> 5) * a.Cout * 4 + (lane & 31) * 16 : lane * 16;
        const int npend = (NOB + k16 - 1) / k16 < 4 ? 4 : (NOB + k16 - 1) / k16;       // pieces per stage: all of them within the next item's stages
        const float oinv = 1.0f / (f16_scale_from_amax(a.amax) * F16_WSCALE);          // undoes the operand scales (a power of two)
        const bool oact = a.stat == nullptr && a.act != VIAI_ACT_NONE;
        f32x4 obias = {0.f, 0.f, 0.f, 0.f};                                            // bias of this lane's four channels in the pending item's block
        auto flush = [&](int cnt) {
            const int hi = pend + cnt;
#pragma unroll
            for (int j = 0; j < NOB; ++j)
                if (j >= pend && j < hi) {
                    f32x4 o = __builtin_bit_cast(f32x4, obuf[j]) * oinv + obias;          // raw accumulators -> conv output
                    if (oact) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = viai_act(o[e], a.act, a.slope);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_out, ovoff, obase[j / (NOB / 2)] + (TN == 1 ? 2 : 1) * (j % (NOB / 2)) * a.Cout * 4, 0);
                }
            pend = hi < NOB ? hi : NOB;
        };
        // start-up: ONLY stage 0 goes out before the first barrier (with two stages queued by every CU of the chip at once the consumers' first barrier came
        // ~9 000 cycles into the kernel: the second batch's issue waits behind the first's back-pressure); stage 1 follows behind barrier 0 together with stage 2
        if (nstage > 0) issue(0);
        auto wait_older = [&](bool younger) {         // everything older than the batch just issued has landed (no batch issued: everything)
            if (!younger) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (NPL == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        };
        wait_older(false);
        for (int q = 0; q < nstage; ++q) {
            S2_STAMP(1, q, 0);
            __syncthreads();                         // stage q landed (every loader waited); the consumers are done with stage q - 1
            S2_STAMP(1, q, 1);
            if (q == 0 && nstage > 1) issue(1);
            if (q + 2 < nstage) issue(q + 2);
            S2_STAMP(1, q, 2);
            wait_older(q + 2 < nstage);              // stage q + 1 landed, q + 2 in flight
            S2_STAMP(1, q, 3);
            // output pieces of the previous item, a few per stage: 64 KB per item and CU in one burst sat in front of the consumers' weight-fragment loads
            // (the stage behind an epilogue took 6 800 cycles instead of 4 400) and kept the loaders from the barrier
            flush(npend);
            if (++kl == k16) {                       // last k-step of an item: its tile arrives through LDS in two halves (see the consumers' epilogue)
                kl = 0;
                flush(NOB);                          // (nothing left unless the item was short)
                int tx, ty, n, nb;
                item_of(q / k16, tx, ty, n, nb);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    __syncthreads();                 // E1 / E3
                    // row lw of the half (tile row 4 hb + lw): 16 pixels x CW channels, contiguous in the hand-off buffer
#pragma unroll
                    for (int j = 0; j < NOB / 2; ++j) obuf[(NOB / 2) * hb + j] = *reinterpret_cast<const u32x4*>(smem_s2 + W::OUT + lw * 16 * CW * 4 + j * 1024 + lane * 16);
                    obase[hb] = __builtin_amdgcn_readfirstlane((((n * g.OH + ty * S2_TH + 4 * hb + lw) * g.OW + tx * S2_TW) * a.Cout + nb * CW) * 4);
                    // (through LDS, written by the consumers: a global load in this wave's queue would make hipcc drain `vmcnt(0)` -- the DMA in flight -- in front of
                    // every output piece that uses it)
                    if (hb == 1 && a.bias != nullptr) obias = *reinterpret_cast<const f32x4*>(smem_s2 + W::BIAS + (lane & (CW / 4 - 1)) * 16);
                    if (hb == 0) __syncthreads();    // E2: half 0 is in registers (the barrier's fence waited for the reads), the consumers may write half 1
                }
                if (lw < (S == 2 ? 2 : 1) && a.stat != nullptr) {   // loader b stores partial block b of the tile: 2 CW floats [mean | M2][CW], lane -> four consecutive, TN rounds
                    // (scalar reads: hipcc turned `u32x4 sv = {}; if (..) sv = *(u32x4*)p; ... sv[i]` into four stores of element 0 -- the
                    // vector-element defect of DESIGN 9.3)
                    const int blk = S == 2 ? (n * (2 * sa.tiles_y) + 2 * ty + lw) * sa.tiles_x + tx : (n * sa.tiles_y + ty) * sa.tiles_x + tx;
#pragma unroll
                    for (int i = 0; i < TN; ++i) {
                        const int idx = (i * 64 + lane) * 4;
                        const float* sp = reinterpret_cast<const float*>(smem_s2 + W::STAT) + lw * 2 * CW + idx;
                        const float sv0 = sp[0], sv1 = sp[1], sv2 = sp[2], sv3 = sp[3];
                        const int which = idx / CW, ch = nb * CW + idx % CW;
                        float* sd = a.stat + (size_t)(which * a.Cout + ch) * a.nblk_m + blk;
                        sd[0] = sv0; sd[a.nblk_m] = sv1; sd[2 * (size_t)a.nblk_m] = sv2; sd[3 * (size_t)a.nblk_m] = sv3;
                    }
                }
                pend = 0;
            }
        }
        flush(NOB);
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumer waves
    // Four waves, one per SIMD: wave wn owns ALL 128 pixels of the tile (four MFMA row tiles) x channels 32 wn .. + 31 of the block.  With eight
    // 64-pixel waves every weight fragment was fetched twice per block: 144 KB per stage through a 64 B/clk vector-memory path = 2 300 of the
    // stage's 3 456 MFMA cycles, and the loaders' DMA queued behind it (tools/probes/s2_dma_bench.hip: loader issue 3 700 - 4 700 cycles per stage).
    const int wn = wave;
    const float ascale = f16_scale_from_amax(a.amax);
    const int NT = a.Cout / 32;
    const int frag_plane = NT * g.wtaps * k16 * 1024;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 2 * frag_plane, 0x00020000);
    int sl[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) sl[t] = __builtin_amdgcn_readfirstlane(sa.slot[t]);
    // A fragment of MFMA row i = lane & 31 of row tile m (tile pixel (2 m + (i >> 4), i & 15)), k-half lane >> 5, window column tx: byte offset in the stage
    const int pc = lane & 15, kh = lane >> 5, rr = (lane & 31) >> 4;
    int abase[3];                                                                 // leading plane (remainder: ^ 32).  S = 2: tx = 0 (px 0, +0), 1 (px 1, +0), 2 (px 0, +1)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
        const int c = S == 2 ? pc + (tx >> 1) : pc + tx, pitch = (S == 2 && (tx & 1)) ? 16 : 20;
        abase[tx] = (rr * pitch + c) * 64 + ((kh ^ ((c >> 2) & 3)) * 16);
    }
    const int half = lane >> 5, col = lane & 31;
    const float inv = 1.0f / (ascale * F16_WSCALE);

    f32x16 acc[S2_TM][TN];
#pragma unroll
    for (int m = 0; m < S2_TM; ++m)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

    // weight fragments of (window position t, k-step kk) of this wave's 32 TN channels: both planes.  NINE sets, one per window position (so the set of a
    // tap is the same in every stage and every register index is static), requested S2_BD taps ahead -- across the stage boundary: the compiler-counted
    // `vmcnt` of a tap then leaves 2 TN (S2_BD - 1) younger loads in flight.  With three sets / two taps ahead hipcc sank every request to within four MFMAs
    // of its use and the stage ran at 5 200 - 7 900 ticks against 3 900 of MFMA issue (tools/probes/s2_dma_bench.hip, -DVIAI_PROF).
#ifndef VIAI_S2_BD
#define VIAI_S2_BD 5
#endif
    constexpr int S2_BD = TN == 1 ? VIAI_S2_BD : 2;
    u32x4 B[9][TN][2];
    int ntb = 0;                                                                  // first channel tile of this wave in the current item
    auto gloadB = [&](u32x4 (&b)[TN][2], int t, int kk_, int ok_, int nt_) {
        const int kk = __builtin_amdgcn_readfirstlane(kk_);
        const int dead = ok_ ? 0 : DP_OOB;
        const int soff = dead ? 0 : (sl[t] * k16 + kk) * 1024;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int voff = ((nt_ + j) * g.wtaps * k16 * 1024 + lane * 16) | dead;
            b[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff, soff, 0);
            b[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff, soff + frag_plane, 0);
        }
    };
    int tx_, ty_, n_, nb_;
    if (nmine > 0) {
        item_of(0, tx_, ty_, n_, nb_); ntb = (nb_ * 4 + wn) * TN;
#pragma unroll
        for (int t = 0; t < S2_BD; ++t) gloadB(B[t], t, 0, 1, ntb);
    }

    int k = 0, kk = 0;                                                            // item, k-step of stage q
    for (int q = 0; q < nstage; ++q) {
        S2_STAMP(0, q, 0);
        __syncthreads();                              // (fence + s_barrier: the fragment reads below must not move above it; the weight prefetch stays in flight)
        S2_STAMP(0, q, 1);
        const unsigned char* Sb = smem_s2 + (q % S2_NSTG) * W::STAGE;
        const bool last = kk + 1 == k16;                                          // last k-step of the item: the next stage's fragments belong to item k + 1
        int ntb_next = ntb;
        if (last && k + 1 < nmine) { int a0, a1, a2, nb2; item_of(k + 1, a0, a1, a2, nb2); ntb_next = (nb2 * 4 + wn) * TN; }
        const int kk_next = last ? 0 : kk + 1, ok_next = q + 1 < nstage;
        // A fragments: ONE set of leading and ONE of remainder pieces per row tile (32 registers; double-buffering both planes would not fit beside the
        // 16 TN accumulator tiles).  The products of a tap run rem x lead-B first, then lead x rem-B and lead x lead-B, so the remainder registers are free
        // after the first 4 TN MFMAs of tap t and are refilled for tap t + 1 under the other 8 TN; the leading registers are refilled behind the tap's last MFMA
        // and are not needed before the 4 TN rem MFMAs of tap t + 1 have run.
        u32x4 alead[S2_TM], arem[S2_TM];
        auto aaddr = [&](int t, int m) -> int {
            constexpr int PB[4] = {S2_P0 * 64, S2_P1 * 64, S2_P2 * 64, S2_P3 * 64};
            const int ty = t / 3, tx = t % 3;
            const int pitch = (S == 2 && (tx & 1)) ? 16 : 20;
            return (S == 2 ? PB[(ty & 1) * 2 + (tx & 1)] + (ty >> 1) * pitch * 64 : ty * pitch * 64) + m * 2 * pitch * 64;
        };
        auto loadLead = [&](int t) {
#pragma unroll
            for (int m = 0; m < S2_TM; ++m) alead[m] = *reinterpret_cast<const u32x4*>(Sb + abase[t % 3] + aaddr(t, m));
        };
        auto loadRem = [&](int t) {
#pragma unroll
            for (int m = 0; m < S2_TM; ++m) arem[m] = *reinterpret_cast<const u32x4*>(Sb + (abase[t % 3] ^ 32) + aaddr(t, m));
        };
        loadRem(0);
        loadLead(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ta = t + S2_BD;
            if (ta < 9) gloadB(B[ta], ta, kk, 1, ntb);
            else gloadB(B[ta - 9], ta - 9, kk_next, ok_next, ntb_next);
            const u32x4 (&b)[TN][2] = B[t];
            // smallest partial products first: rem x lead, lead x rem, lead x lead (the order of conv_halo_wide_f16_kernel)
#pragma unroll
            for (int m = 0; m < S2_TM; ++m)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, arem[m]), __builtin_bit_cast(f16x8, b[j][0]), acc[m][j], 0, 0, 0);
            // one consumer wave per SIMD: the weight-fragment requests go BETWEEN these MFMAs, not in a block in front of them
#pragma unroll
            for (int i = 0; i < 2 * TN; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TN - 2 * TN, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) loadRem(t + 1);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int m = 0; m < S2_TM; ++m)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, alead[m]), __builtin_bit_cast(f16x8, b[j][1 - pr]), acc[m][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * TN - 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) loadLead(t + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        S2_STAMP(0, q, 2);
        if (last) {
            // ------------------------------------------------------------------------------------------ epilogue of item k
            // The consumers do NOT store to global memory: a store in this wave's queue would sit between its weight-fragment loads, hipcc would have to
            // drain `vmcnt(0)` behind it (mixed loads and stores may retire out of order), and the 64 dword stores + that drain measured 6 500 of an item's
            // 24 000 cycles.  The tile goes to the LOADER waves through LDS in two halves of 64 pixels x 128 channels (32 KB, rows 0 - 3 then 4 - 7)
            // together with its BatchNorm partials; they store it as 16-byte pieces from their own queue while this wave runs the next item.
            item_of(k, tx_, ty_, n_, nb_);
            // (the accumulators go out RAW: the loaders apply the operand-scale factor, the bias and a fused activation to the 16-byte pieces they store;
            // the statistics are taken on the raw values and scaled at the end -- the factor is a power of two, so this is exact)
            float* ob = reinterpret_cast<float*>(smem_s2 + W::OUT) + (4 * half) * CW + wn * 32 * TN + col;
            float* sb = reinterpret_cast<float*>(smem_s2 + W::STAT) + wn * 32 * TN + col;
            auto put = [&](int hb) {                  // pixel (2 (m & 1) + (e >> 3), (e & 3) + 8 ((e >> 2) & 1) + 4 half) of the half
#pragma unroll
                for (int m = 2 * hb; m < 2 * hb + 2; ++m)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) ob[((2 * (m & 1) + (e >> 3)) * 16 + (e & 3) + 8 * ((e >> 2) & 1)) * CW + j * 32] = acc[m][j][e];
            };
            put(0);
            __syncthreads();                          // E1: half 0 is in LDS; the loaders read it while this wave reduces its statistics
            constexpr int NBLK = S == 2 ? 2 : 1, MPB = S2_TM / NBLK;      // partial blocks per tile and row tiles per block
            float mw[NBLK][TN], m2[NBLK][TN];
            if (a.stat != nullptr) {
                // (mean, M2) per partial block and channel.  Stride 2: two 4 x 16 pixel blocks per tile (the partial-block geometry of the register-staged kernel's
                // 64-pixel tiles, viai_halo_s2_rows = 4); stride 1: the 8 x 16 tile.  Two-pass; one wave per SIMD has no neighbour to hide a long dependent add
                // chain behind, so four partial sums
#pragma unroll
                for (int hb = 0; hb < NBLK; ++hb)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int m = MPB * hb; m < MPB * hb + MPB; ++m)
#pragma unroll
                            for (int e = 0; e < 16; ++e) t[e & 3] += acc[m][j][e];
                        float ts = (t[0] + t[1]) + (t[2] + t[3]);
                        ts += __shfl_xor(ts, 32, 64);
                        const float mraw = ts / (float)(32 * MPB);
                        float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int m = MPB * hb; m < MPB * hb + MPB; ++m)
#pragma unroll
                            for (int e = 0; e < 16; ++e) { const float d = acc[m][j][e] - mraw; u[e & 3] += d * d; }
                        float us = (u[0] + u[1]) + (u[2] + u[3]);
                        us += __shfl_xor(us, 32, 64);
                        mw[hb][j] = mraw * inv + (a.bias != nullptr ? a.bias[nb_ * CW + (wn * TN + j) * 32 + col] : 0.f);
                        m2[hb][j] = us * (inv * inv);
                    }
            }
            __syncthreads();                          // E2: the loaders hold half 0 in registers
            put(1);
            if (a.bias != nullptr && half == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j) reinterpret_cast<float*>(smem_s2 + W::BIAS)[(wn * TN + j) * 32 + col] = a.bias[nb_ * CW + (wn * TN + j) * 32 + col];
            }
            if (a.stat != nullptr && half == 0) {
#pragma unroll
                for (int hb = 0; hb < NBLK; ++hb)
#pragma unroll
                    for (int j = 0; j < TN; ++j) { sb[hb * 2 * CW + j * 32] = mw[hb][j]; sb[hb * 2 * CW + CW + j * 32] = m2[hb][j]; }
            }
            __syncthreads();                          // E3: half 1 and the tile's partials are in LDS
#pragma unroll
            for (int m = 0; m < S2_TM; ++m)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;
            ntb = ntb_next;
            ++k; kk = 0;
        } else ++kk;
        S2_STAMP(0, q, 3);
    }
#ifdef VIAI_PROF
    if (tid == 0) sa.prof[(((size_t)blockIdx.x * 2 + 1) * 32 + 31) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The loader / consumer kernel over LINEAR pixel tiles (round 5): stride-1 3 x 3 pad-1 layers on maps that are not whole 8 x 16 tiles -- ResNet-18's
// 56 / 28 / 14 / 7-pixel maps (networks/ResNet.py:26-55: conv3x3 -> bn -> relu -> conv3x3 -> bn, both directions; Image_Embedding.py:187-200).
//
// A work item is PIX = 128 PW CONSECUTIVE pixels of the flattened [N][H][W] map x CW = 256 / PW channels; the four consumer waves split it PW ways along
// the pixels and 4 / PW ways along the channels (128 pixels x 64 channels each, as in conv_wide_dma_kernel<1, 2>).  Window position (dy, dx) of pixel m is
// pixel m + dy W + dx of the same flattened map, so a stage holds the k-step's records of pixels m0 - W - 1 .. m0 + PIX + W (one contiguous range of the
// P16 tensor: the loaders' offsets are linear in the record index, no division anywhere) and a fragment read is 32 consecutive 64-byte records
// shifted by the tap.  What the 2-D patch did with out-of-range lanes -- the zero padding -- the CONSUMERS do here: a lane whose window position falls
// outside its image (nine validity bits per pixel, worked out once per item) reads the all-zero bank row at LDS byte 0 instead.
// Records keep the piece swizzle of the 2-D kernels, position = piece ^ ((record >> 2) & 3): a shift by a whole number of records moves a read to other
// records, the 16 lanes of a `ds_read_b128` group still take 16 consecutive ones, four per bank-row quarter.
template <int PW>
struct LinDma {
    static constexpr int NWN = 4 / PW;                                 // consumer waves along the channels
    static constexpr int CW = 64 * NWN;                                // channels of an item
    static constexpr int PIX = 128 * PW;                               // pixels of an item
    static constexpr int NPL = PW == 4 ? 10 : PW == 2 ? 6 : 4;         // DMA instructions per loader wave and stage (16 records each)
    static constexpr int NREC = 64 * NPL;                              // record slots of a stage >= PIX + 2 W + 2: W <= 63
    static constexpr int STAGE = NREC * 64;
    static constexpr int RING = 1024;                                  // behind the zero record
    static constexpr int NH = PW == 4 ? 4 : 2;                         // hand-off parts of the item's 128 KB output
    static constexpr int OUT = RING + S2_NSTG * STAGE, OUT_BYTES = PIX * CW * 4 / NH;
    static constexpr int STAT = OUT + OUT_BYTES;                       // [PW 128-pixel blocks][mean | M2][CW]
    static constexpr int BIAS = STAT + PW * 2 * CW * 4;
    static constexpr int LDS = BIAS + CW * 4;
    static constexpr int NOB = 32;                                     // 1 KB output pieces per loader wave and item
    static constexpr int NOP = NOB / NH;                               // ... per part
};

struct LinArgs {
    int H, W, HW, M;                                                   // map, pixels per image, pixels in all
    float rhw; unsigned mw;                                            // 1 / HW (float quotient, corrected), division magic of W
    int nnb; unsigned mnb;                                             // channel blocks per pixel tile (item = tile * nnb + nb)
    int nitems;
    int slot[9];                                                       // weight slot of window position (row * 3 + col)
    int stat_parts;                                                    // > 0 (round 6): BatchNorm partials merged over a block's items, [2][Cout][stat_parts] (viai_lin_dma_stat_merge)
#ifdef VIAI_PROF
    unsigned long long* prof;                                          // [block][2 roles][32 stages][4 stamps] (tools/probes/s2_dma_bench.hip)
#endif
};

__device__ __forceinline__ void dma_batch(const i32x4& rs, unsigned lds0, int soff, const int (&v)[6]) { dma_batch6(rs, lds0, soff, v[0], v[1], v[2], v[3], v[4], v[5]); }

template <int PW>
__global__ __launch_bounds__(S2_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_lin_dma_kernel(const ConvArgs a, const LinArgs sa) {
    using L = LinDma<PW>;
    constexpr int NPL = L::NPL, CW = L::CW, NOB = L::NOB, NH = L::NH, NOP = L::NOP, NWN = L::NWN, PIX = L::PIX, TN = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_ln[];     // [zero record] [3 stages] [output part] [partials] [bias]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.C1, k16 = Cin / 16;
    const int G = gridDim.x;
    const int nmine = (sa.nitems - (int)blockIdx.x + G - 1) / G;
    const int nstage = nmine * k16;
    const int nrec = PIX + 2 * sa.W + 2;
    auto item_of = [&](int k, int& m0, int& nb) {
        const int item = xcd_remap(blockIdx.x + k * G, sa.nitems);
        const int tile = div_magic(item, sa.mnb);
        nb = item - tile * sa.nnb;
        m0 = tile * PIX;
    };
    if (tid < 64) reinterpret_cast<unsigned*>(smem_ln)[tid] = 0u;               // the zero row: 256 bytes (visible behind the first stage barrier)

    if (wave >= S2_NCONS) {
        // ------------------------------------------------------------------------------------------------ loader waves
        const int lw = wave - S2_NCONS;
        const i32x4 rs_in = rsrc_sgpr(a.in, (unsigned)((long)sa.M * Cin * 4));
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_ln + L::RING + lw * 1024;
        // lane -> (record 16 i + (lane >> 2), position lane & 3) of instruction i = lw + 4 j
        int vint[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int rec = 16 * (lw + 4 * j) + (lane >> 2);
            const int qp = (lane & 3) ^ ((rec >> 2) & 3);                         // piece held at this position: plane qp >> 1, k-half qp & 1
            vint[j] = rec < nrec ? rec * Cin * 4 + (qp >> 1) * 64 + (qp & 1) * 16 : DP_OOB;
        }
        auto issue = [&](int q) {                    // stage q = (item q / k16, k-step q % k16) -> ring slot q % 3
            const int k = q / k16, kk = q - k * k16;
            int m0, nb;
            item_of(k, m0, nb);
            const int ms = m0 - sa.W - 1;                                         // pixel of record 0
            const unsigned lds0 = lds_base + (q % S2_NSTG) * L::STAGE;
            const int koff = (kk >> 1) * 128 + (kk & 1) * 32;                     // chunk, k-step within the chunk's 128-byte record
            if (ms >= 0 && ms + nrec <= sa.M) {
                const int soff = __builtin_amdgcn_readfirstlane(ms * Cin * 4 + koff);
                dma_batch(rs_in, lds0, soff, vint);
            } else {                                  // first / last pixels of the tensor: records in front of pixel 0 / behind pixel M - 1 go out of range (zeros)
                int v[NPL];
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const int mm = ms + 16 * (lw + 4 * j) + (lane >> 2);
                    v[j] = (vint[j] == DP_OOB || mm < 0 || mm >= sa.M) ? DP_OOB : vint[j] + ms * Cin * 4;
                }
                dma_batch(rs_in, lds0, koff, v);
            }
        };
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)((long)sa.M * a.Cout * 4), 0x00020000);
        int kl = 0;
        u32x4 obuf[NOB];                             // this wave's share of the last finished item: NH parts x NOP pieces (PW pixels x CW channels each)
        int obase[NH], pend = NOB;
#pragma unroll
        for (int h = 0; h < NH; ++h) obase[h] = 0;
        const int ovoff = (lane / (CW / 4)) * a.Cout * 4 + (lane % (CW / 4)) * 16;
        const int npend = (NOB + k16 - 1) / k16 < 4 ? 4 : (NOB + k16 - 1) / k16;
        const float oinv = 1.0f / (f16_scale_from_amax(a.amax) * F16_WSCALE);
        const bool oact = a.stat == nullptr && a.act != VIAI_ACT_NONE;
        f32x4 obias = {0.f, 0.f, 0.f, 0.f};
        auto flush = [&](int cnt) {
            const int hi = pend + cnt;
#pragma unroll
            for (int j = 0; j < NOB; ++j)
                if (j >= pend && j < hi) {
                    f32x4 o = __builtin_bit_cast(f32x4, obuf[j]) * oinv + obias;
                    if (oact) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = viai_act(o[e], a.act, a.slope);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_out, ovoff, obase[j / NOP] + (j % NOP) * PW * a.Cout * 4, 0);
                }
            pend = hi < NOB ? hi : NOB;
        };
        if (nstage > 0) issue(0);
        auto wait_older = [&](bool younger) {
            if (!younger) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (NPL == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if constexpr (NPL == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        };
        wait_older(false);
        // the part's pixels this wave stores: part-local pixels lw Q .. + Q - 1 (Q = PIX / NH / 4); part-local pixel (wp (4 / NH) + ml) 32 + i is item pixel
        // wp 128 + (h (4 / NH) + ml) 32 + i
        constexpr int Q = PIX / NH / 4, PPW = 128 / NH;
        const int lpix = (lw * Q / PPW) * 128 + (lw * Q) % PPW;
        for (int q = 0; q < nstage; ++q) {
            S2_STAMP(1, q, 0);
            __syncthreads();                         // stage q landed (every loader waited); the consumers are done with stage q - 1
            S2_STAMP(1, q, 1);
            if (q == 0 && nstage > 1) issue(1);
            if (q + 2 < nstage) issue(q + 2);
            S2_STAMP(1, q, 2);
            wait_older(q + 2 < nstage);
            S2_STAMP(1, q, 3);
            flush(npend);
            if (++kl == k16) {                       // last k-step of an item: its output arrives through LDS in NH parts
                kl = 0;
                flush(NOB);
                int m0, nb;
                item_of(q / k16, m0, nb);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    __syncthreads();                 // part h is in LDS
#pragma unroll
                    for (int j = 0; j < NOP; ++j) obuf[NOP * h + j] = *reinterpret_cast<const u32x4*>(smem_ln + L::OUT + (lw * NOP + j) * 1024 + lane * 16);
                    obase[h] = __builtin_amdgcn_readfirstlane(((m0 + lpix + h * PPW) * a.Cout + nb * CW) * 4);
                    if (h == NH - 1 && a.bias != nullptr) obias = *reinterpret_cast<const f32x4*>(smem_ln + L::BIAS + (lane & (CW / 4 - 1)) * 16);
                    if (h < NH - 1) __syncthreads(); // part h is in registers, the consumers may write part h + 1
                }
                if (lw < PW && a.stat != nullptr && sa.stat_parts == 0) {  // loader b stores the partials of 128-pixel block b of the item: [mean | M2][CW]
                    const int blk = m0 / 128 + lw;
#pragma unroll
                    for (int i = 0; i < (2 * CW + 255) / 256; ++i) {
                        const int idx = (i * 64 + lane) * 4;
                        if (idx < 2 * CW) {
                            const float* sp = reinterpret_cast<const float*>(smem_ln + L::STAT) + lw * 2 * CW + idx;
                            const float sv0 = sp[0], sv1 = sp[1], sv2 = sp[2], sv3 = sp[3];
                            const int which = idx / CW, ch = nb * CW + idx % CW;
                            float* sd = a.stat + (size_t)(which * a.Cout + ch) * a.nblk_m + blk;
                            sd[0] = sv0; sd[a.nblk_m] = sv1; sd[2 * (size_t)a.nblk_m] = sv2; sd[3 * (size_t)a.nblk_m] = sv3;
                        }
                    }
                }
                pend = 0;
            }
        }
        flush(NOB);
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumer waves
    const int wp = wave / NWN, wn = wave % NWN;
    const float ascale = f16_scale_from_amax(a.amax);
    const int NT = a.Cout / 32;
    const int wtaps = a.g.wtaps;
    const int frag_plane = NT * wtaps * k16 * 1024;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 2 * frag_plane, 0x00020000);
    int sl[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) sl[t] = __builtin_amdgcn_readfirstlane(sa.slot[t]);
    // A fragment of MFMA row i = lane & 31 of row tile m (item pixel wp 128 + 32 m + i), k-half lane >> 5, window position t: leading piece (remainder: ^ 32)
    const int half = lane >> 5, col = lane & 31;
    int at[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int rec = wp * 128 + col + (t / 3) * sa.W + (t % 3);
        at[t] = L::RING + rec * 64 + ((half ^ ((rec >> 2) & 3)) * 16);
    }
    const float inv = 1.0f / (ascale * F16_WSCALE);

    f32x16 acc[S2_TM][TN];
#pragma unroll
    for (int m = 0; m < S2_TM; ++m)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

    constexpr int BD = 2;
    u32x4 B[9][TN][2];
    int ntb = 0;
    auto gloadB = [&](u32x4 (&b)[TN][2], int t, int kk_, int ok_, int nt_) {
        const int kk = __builtin_amdgcn_readfirstlane(kk_);
        const int dead = ok_ ? 0 : DP_OOB;
        const int soff = dead ? 0 : (sl[t] * k16 + kk) * 1024;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int voff = ((nt_ + j) * wtaps * k16 * 1024 + lane * 16) | dead;
            b[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff, soff, 0);
            b[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff, soff + frag_plane, 0);
        }
    };
    // validity of the nine window positions of this lane's pixel in row tile m (bit t): positions outside the pixel's image read the zero record
    int vm[S2_TM];
    auto masks_of = [&](int m0) {
#pragma unroll
        for (int m = 0; m < S2_TM; ++m) {
            const int p = m0 + wp * 128 + m * 32 + col;
            int n = (int)((float)p * sa.rhw);                                      // p < 2^24: the float quotient is within one of the integer one
            int r = p - n * sa.HW;
            if (r < 0) r += sa.HW; else if (r >= sa.HW) r -= sa.HW;
            const int y = div_magic(r, sa.mw), x = r - y * sa.W;
            int v = 0x1ff;
            if (y == 0) v &= ~0x007;
            if (y == sa.H - 1) v &= ~0x1c0;
            if (x == 0) v &= ~0x049;
            if (x == sa.W - 1) v &= ~0x124;
            vm[m] = v;
        }
    };
    // BatchNorm partials merged over this block's items (stat_parts > 0: one channel block per layer, so a wave's channels never change): Chan's update of
    // (count, mean, M2) per channel and wave -- the 25 088 partial blocks per channel of ResNet layer1 on 1024 frames become 1024, and bn_finalize's launch, which
    // waited 238 us as run behind the other chain's resident blocks, shrinks to the small form
    float rcnt = 0.f, rmean[TN], rm2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { rmean[j] = 0.f; rm2[j] = 0.f; }
    int m0_, nb_;
    if (nmine > 0) {
        item_of(0, m0_, nb_); ntb = (nb_ * NWN + wn) * TN;
        masks_of(m0_);
#pragma unroll
        for (int t = 0; t < BD; ++t) gloadB(B[t], t, 0, 1, ntb);
    }

    int k = 0, kk = 0;
    for (int q = 0; q < nstage; ++q) {
        S2_STAMP(0, q, 0);
        __syncthreads();
        S2_STAMP(0, q, 1);
        const int sbase = (q % S2_NSTG) * L::STAGE;
        const bool last = kk + 1 == k16;
        int ntb_next = ntb, m0_next = m0_;
        if (last && k + 1 < nmine) { int nb2; item_of(k + 1, m0_next, nb2); ntb_next = (nb2 * NWN + wn) * TN; }
        const int kk_next = last ? 0 : kk + 1, ok_next = q + 1 < nstage;
        u32x4 alead[S2_TM], arem[S2_TM];
        // (a masked lane keeps the low 8 address bits -- its place in the 256-byte bank row -- and reads the zero ROW at LDS byte 0: with one zero record for all of
        // them the masked lanes of a 16-lane group landed on banks a valid lane of the group was reading: SQ_LDS_BANK_CONFLICT 0.22 on the 28-pixel maps)
        auto aaddr = [&](int t, int m) -> int { const int ad = at[t] + sbase + m * 2048; return ((vm[m] >> t) & 1) ? ad : (ad & 255); };
        auto loadLead = [&](int t) {
#pragma unroll
            for (int m = 0; m < S2_TM; ++m) alead[m] = *reinterpret_cast<const u32x4*>(smem_ln + aaddr(t, m));
        };
        auto loadRem = [&](int t) {
#pragma unroll
            for (int m = 0; m < S2_TM; ++m) arem[m] = *reinterpret_cast<const u32x4*>(smem_ln + (aaddr(t, m) ^ 32));
        };
        loadRem(0);
        loadLead(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ta = t + BD;
            if (ta < 9) gloadB(B[ta], ta, kk, 1, ntb);
            else gloadB(B[ta - 9], ta - 9, kk_next, ok_next, ntb_next);
            const u32x4 (&b)[TN][2] = B[t];
#pragma unroll
            for (int m = 0; m < S2_TM; ++m)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, arem[m]), __builtin_bit_cast(f16x8, b[j][0]), acc[m][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2 * TN; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TN - 2 * TN, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) loadRem(t + 1);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int m = 0; m < S2_TM; ++m)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, alead[m]), __builtin_bit_cast(f16x8, b[j][1 - pr]), acc[m][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * TN - 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) loadLead(t + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        S2_STAMP(0, q, 2);
        if (last) {
            // ------------------------------------------------------------------------------------------ epilogue of item k (see conv_wide_dma_kernel)
            item_of(k, m0_, nb_);
            constexpr int MPP = S2_TM / NH;                                        // row tiles per part
            // part h holds row tiles h MPP .. + MPP - 1 of every consumer: part-local pixel ((wp MPP + ml) 32 + i), CW channels each
            float* ob = reinterpret_cast<float*>(smem_ln + L::OUT) + (wp * MPP * 32 + 4 * half) * CW + wn * 64 + col;
            float* sb = reinterpret_cast<float*>(smem_ln + L::STAT) + wp * 2 * CW + wn * 64 + col;
            auto put = [&](int h) {                   // MFMA row (e & 3) + 8 (e >> 2) + 4 half of row tile m
#pragma unroll
                for (int m = h * MPP; m < h * MPP + MPP; ++m)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) ob[((m - h * MPP) * 32 + (e & 3) + 8 * (e >> 2)) * CW + j * 32] = acc[m][j][e];
            };
            float mw[TN], m2[TN];
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                put(h);
                if (h == NH - 1) {
                    if (a.bias != nullptr && half == 0 && wp == 0) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) reinterpret_cast<float*>(smem_ln + L::BIAS)[(wn * TN + j) * 32 + col] = a.bias[nb_ * CW + (wn * TN + j) * 32 + col];
                    }
                    if (a.stat != nullptr && half == 0 && sa.stat_parts == 0) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) { sb[j * 32] = mw[j]; sb[CW + j * 32] = m2[j]; }
                    }
                }
                __syncthreads();                      // part h is in LDS
                if (h == 0 && a.stat != nullptr) {
                    // (mean, M2) of this wave's 128 pixels per channel, on the raw accumulators (the scale is a power of two: exact); two-pass, four partial sums
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int m = 0; m < S2_TM; ++m)
#pragma unroll
                            for (int e = 0; e < 16; ++e) t4[e & 3] += acc[m][j][e];
                        float ts = (t4[0] + t4[1]) + (t4[2] + t4[3]);
                        ts += __shfl_xor(ts, 32, 64);
                        const float mraw = ts / 128.f;
                        float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int m = 0; m < S2_TM; ++m)
#pragma unroll
                            for (int e = 0; e < 16; ++e) { const float d = acc[m][j][e] - mraw; u[e & 3] += d * d; }
                        float us = (u[0] + u[1]) + (u[2] + u[3]);
                        us += __shfl_xor(us, 32, 64);
                        mw[j] = mraw * inv + (a.bias != nullptr ? a.bias[nb_ * CW + (wn * TN + j) * 32 + col] : 0.f);
                        m2[j] = us * (inv * inv);
                    }
                    if (sa.stat_parts > 0) {
                        const float nn = rcnt + 128.f, f = 128.f / nn;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float dl = mw[j] - rmean[j];
                            rmean[j] += dl * f;
                            rm2[j] += m2[j] + dl * dl * (rcnt * f);
                        }
                        rcnt = nn;
                    }
                }
                if (h < NH - 1) __syncthreads();      // the loaders hold part h in registers
            }
#pragma unroll
            for (int m = 0; m < S2_TM; ++m)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;
            ntb = ntb_next;
            ++k; kk = 0;
            if (k < nmine) { m0_ = m0_next; masks_of(m0_); }
        } else ++kk;
        S2_STAMP(0, q, 3);
    }
    if (a.stat != nullptr && sa.stat_parts > 0 && half == 0 && nmine > 0) {
        const int part = blockIdx.x * PW + wp;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ch = (wn * TN + j) * 32 + col;
            a.stat[(size_t)ch * sa.stat_parts + part] = rmean[j];
            a.stat[(size_t)(a.Cout + ch) * sa.stat_parts + part] = rm2[j];
        }
    }
}

}  // namespace

// stride-2 3 x 3 pad-1 forward layers the producer / consumer kernel takes: P16 input with Cin a multiple of 32 (one source), Cout a multiple
// of 128 into one destination, the input exactly twice the output, whole 8 x 16 tiles, 32-bit byte offsets
static bool wide_dma_common(const ConvArgs& a) {
    const ConvGeom& g = a.g;
    if (!viai_halo_dma_on() || !a.in_p16 || a.amax == nullptr || a.C2 != 0 || a.C1 % 32 != 0 || a.Cout % 128 != 0 || a.OC1 != a.Cout) return false;
    if (g.ntaps != 9 || g.OH % S2_TH != 0 || g.OW % S2_TW != 0 || g.ly != 1 || g.lx != 1 || g.SH != g.OH || g.SW != g.OW) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        if (g.dy[t] < -1 || g.dy[t] > 1 || g.dx[t] < -1 || g.dx[t] > 1) return false;
        seen |= 1u << ((g.dy[t] + 1) * 3 + (g.dx[t] + 1));
    }
    if (seen != 0x1ffu) return false;
    if ((long)g.N * g.IH * g.IW * a.C1 * 4 >= (1l << 31) || (long)g.N * g.OH * g.OW * a.Cout * 4 >= (1l << 31)) return false;
    const long items = (long)g.N * (g.OH / S2_TH) * (g.OW / S2_TW) * (a.Cout / 128);
    return items * 64 < (1l << 31);
}
bool viai_conv_s2_dma_ok(const ConvArgs& a) {
    const ConvGeom& g = a.g;
    return wide_dma_common(a) && g.my == 2 && g.mx == 2 && g.IH == 2 * g.OH && g.IW == 2 * g.OW;
}
// stride-1 3 x 3 pad-1 layers (forward, and the data gradient as the forward of the flipped filter) with 256 k output channels and at least one
// 128-pixel x 256-channel work item per CU: D.conv3 in both directions (networks/Discriminator_Networks.py:29-31)
bool viai_conv_s1_dma_ok(const ConvArgs& a) {
    const ConvGeom& g = a.g;
    if (!wide_dma_common(a) || g.my != 1 || g.mx != 1 || g.IH != g.OH || g.IW != g.OW || a.Cout % 256 != 0) return false;
    return (long)g.N * (g.OH / S2_TH) * (g.OW / S2_TW) * (a.Cout / 256) >= 256;
}

template <int S, int TN>
static int launch_wide_dma(ConvArgs& a, hipStream_t st) {
    using W = WideDma<S, TN>;
    const ConvGeom& g = a.g;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wide_dma_kernel<S, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, W::LDS);
        attr_done = true;
    }
    S2Args sa;
    sa.tiles_x = g.OW / S2_TW; sa.tiles_y = g.OH / S2_TH; sa.mx = tile_magic(sa.tiles_x); sa.my = tile_magic(sa.tiles_y);
    sa.nnb = a.Cout / W::CW; sa.mnb = tile_magic(sa.nnb);
    sa.nitems = g.N * sa.tiles_y * sa.tiles_x * sa.nnb;
    for (int t = 0; t < 9; ++t) sa.slot[(g.dy[t] + 1) * 3 + (g.dx[t] + 1)] = g.ws[t];
#ifdef VIAI_PROF
    sa.prof = viai_dma_prof_buf;
#endif
    a.nblk_m = a.M / (S == 2 ? 64 : 128);                     // BatchNorm partial blocks: 4 x 16 pixels (stride 2) / the 8 x 16 tile (stride 1)
    a.nblk_n = sa.nnb;
    int grid = viai_dma_grid();
    if (grid > sa.nitems) grid = sa.nitems;
    VIAI_LAUNCH((conv_wide_dma_kernel<S, TN>), dim3(grid), dim3(S2_THREADS), W::LDS, st, a, sa);
    return viai_launch_status();
}

int viai_conv_s2_dma_launch(ConvArgs& a, hipStream_t st) {
    if (!viai_conv_s2_dma_ok(a)) return (int)hipErrorInvalidValue;
    return launch_wide_dma<2, 1>(a, st);
}
int viai_conv_s1_dma_launch(ConvArgs& a, hipStream_t st) {
    if (!viai_conv_s1_dma_ok(a)) return (int)hipErrorInvalidValue;
    return launch_wide_dma<1, 2>(a, st);
}

// stride-1 3 x 3 pad-1 layers the linear-tile kernel takes (geometry only: the launch also needs the pre-split input and its scale): one source with
// Cin a multiple of 32, Cout a multiple of 64 into one destination, maps up to 63 pixels wide, whole items, at least one item per CU, 32-bit offsets.
// Maps of whole 8 x 16 tiles with 256 k channels stay on conv_wide_dma_kernel<1, 2> (D.conv3: no validity selects in its fragment reads).
static int lin_dma_pw(int Cout) { return Cout % 256 == 0 ? 1 : Cout % 128 == 0 ? 2 : 4; }
bool viai_conv_lin_dma_geom_ok(const ConvArgs& a) {
    const ConvGeom& g = a.g;
    if (!viai_halo_dma_on() || a.C2 != 0 || a.C1 % 32 != 0 || a.Cout % 64 != 0 || a.OC1 != a.Cout) return false;
    if (g.ntaps != 9 || g.ly != 1 || g.lx != 1 || g.my != 1 || g.mx != 1 || g.SH != g.OH || g.SW != g.OW || g.IH != g.OH || g.IW != g.OW || g.OW > 63) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        if (g.dy[t] < -1 || g.dy[t] > 1 || g.dx[t] < -1 || g.dx[t] > 1) return false;
        seen |= 1u << ((g.dy[t] + 1) * 3 + (g.dx[t] + 1));
    }
    if (seen != 0x1ffu) return false;
    const long M = (long)g.N * g.OH * g.OW;
    const int pw = lin_dma_pw(a.Cout);
    if (M % (128 * pw) != 0 || M * a.C1 * 4 >= (1l << 31) || M * a.Cout * 4 >= (1l << 31)) return false;
    if (g.OH % S2_TH == 0 && g.OW % S2_TW == 0 && a.Cout % 256 == 0) return false;
    return (M / (128 * pw)) * (a.Cout / (256 / pw)) >= 256;
}
bool viai_conv_lin_dma_ok(const ConvArgs& a) { return a.in_p16 && a.amax != nullptr && viai_conv_lin_dma_geom_ok(a); }

// BatchNorm partials of the linear-tile kernel merged per persistent block and consumer wave: layers with ONE channel block (Cout = 64 / 128 / 256), where a
// wave's channels are the same for every item.  parts = blocks x PW; part p = (block p / PW, pixel sub-block p % PW) covers 128 x (items of that block) pixels
// (viai_bn_finalize_lin derives the counts from the same numbers).  Returns 0 where the partials stay per 128 pixels.
int viai_lin_dma_stat_merge(long M, int Cout, int* grid, int* pw, int* nitems) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("VIAI_LIN_STAT_MERGE"); on = e ? atoi(e) : 1; }      // (A/B switch: 0 keeps the partials per 128 pixels)
    const int PW = lin_dma_pw(Cout);
    if (!on || Cout != 256 / PW || M % (128 * PW) != 0) return 0;
    const long items = M / (128 * PW);
    if (items >= (1l << 30)) return 0;
    int g = viai_dma_grid();
    if (g > items) g = (int)items;
    if (grid) *grid = g;
    if (pw) *pw = PW;
    if (nitems) *nitems = (int)items;
    return g * PW;
}

template <int PW>
static int launch_lin_dma(ConvArgs& a, hipStream_t st) {
    using L = LinDma<PW>;
    const ConvGeom& g = a.g;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lin_dma_kernel<PW>), hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS);
        attr_done = true;
    }
    LinArgs sa;
    sa.H = g.OH; sa.W = g.OW; sa.HW = g.OH * g.OW; sa.M = g.N * sa.HW;
    sa.rhw = 1.0f / (float)sa.HW; sa.mw = tile_magic(sa.W);
    sa.nnb = a.Cout / L::CW; sa.mnb = tile_magic(sa.nnb);
    sa.nitems = sa.M / L::PIX * sa.nnb;
    for (int t = 0; t < 9; ++t) sa.slot[(g.dy[t] + 1) * 3 + (g.dx[t] + 1)] = g.ws[t];
#ifdef VIAI_PROF
    sa.prof = viai_dma_prof_buf;
#endif
    a.nblk_m = sa.M / 128;                                    // BatchNorm partial blocks: 128 consecutive pixels (viai_bn_finalize with rows = 128)
    a.nblk_n = sa.nnb;
    int grid = viai_dma_grid();
    if (grid > sa.nitems) grid = sa.nitems;
    sa.stat_parts = a.stat != nullptr ? viai_lin_dma_stat_merge(sa.M, a.Cout, nullptr, nullptr, nullptr) : 0;      // ... or merged per block (viai_bn_finalize_lin)
    VIAI_LAUNCH((conv_lin_dma_kernel<PW>), dim3(grid), dim3(S2_THREADS), L::LDS, st, a, sa);
    return viai_launch_status();
}
int viai_conv_lin_dma_launch(ConvArgs& a, hipStream_t st) {
    if (!viai_conv_lin_dma_ok(a)) return (int)hipErrorInvalidValue;
    const int pw = lin_dma_pw(a.Cout);
    return pw == 1 ? launch_lin_dma<1>(a, st) : pw == 2 ? launch_lin_dma<2>(a, st) : launch_lin_dma<4>(a, st);
}
