// seam_pw.hip -- pointwise (1x1, stride 1) convolution with a SHORT reduction (C <= 256) on the gfx950 fp32 matrix cores:
// the ResNet bottleneck expansions / reductions of layer1-3, the FPN laterals of the fine levels, the mask head's ConvTranspose
// (as 1x1 -> 4 sub-pixel groups).  Exact fp32 (v_mfma_f32_32x32x2_f32), same contract as seam_conv2d_f32 on those shapes.
//
// Why a second GEMM kernel.  What tools/mfma_shadow_probe.hip measured on MI355X (profiles/r04_mfma_shadow_probe.txt):
//   * in ONE wave nothing of the vector ALU overlaps an fp32 MFMA: a group of k VALU instructions between two MFMAs costs
//     ~17 + 4.5 k cycles of matrix-pipe idle time; LDS reads, global loads and SALU issued between MFMAs are free;
//   * ANOTHER wave of the same SIMD runs VALU / LDS / memory instructions beside a wave that streams MFMAs without slowing it
//     (64.0 cycles per MFMA either way).
// The implicit GEMM (seam_conv.hip) walks tiles in lock step: the four waves of a block meet at a barrier per 32-k chunk and all
// of them leave the matrix pipe for the epilogue at once (~300 instructions, the residual's HBM latency, 64 KB of stores per
// tile); on a 2-8 chunk reduction that phase is as long as the K loop, and only the statistically offset second block of the CU
// covers it (62-112 TFLOP/s on these layers, profiles/r02_breakdown_f32.txt).
//
// Here the weights are STATIONARY and the waves are INDEPENDENT:
//   * a block (8 waves, one per CU) copies its slab of the weight matrix -- NS = 32 NT output channels x C, <= 133 KB -- into LDS
//     once and then never meets a barrier again;
//   * every wave owns whole pixel rows: a wave tile is 32 MT pixels x NS channels (128 accumulators).  Its A fragments come
//     straight from global memory in MFMA layout (lane = pixel row, 16 bytes = 4 consecutive k; a 4-deep ring, so a load is
//     issued 3-4 k-steps -- ~3 us -- before its MFMAs), its B fragments from the slab with ds_read_b128 (free next to MFMAs),
//     and the K phase contains no vector-ALU instruction at all: offsets are a per-lane constant + SGPR + immediate;
//   * the epilogue (shift, residual / nearest-upsampled residual, ReLU) goes through a wave-private 2 KB LDS transpose, half an
//     MFMA tile at a time, so that residual loads and stores are 128-byte row pieces; it runs while the SIMD's other wave --
//     which is at an unrelated point of its own tile -- keeps the matrix pipe busy.
// Tiles are dealt XCD-aware: row tile t lives on XCD t mod 8, and the K / NS slab groups of one XCD walk the same row tiles at
// about the same time, so an activation row is read from HBM once and from that XCD's L2 by the other slabs.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include "seam_opts.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

// LDS access by 32-bit byte address (the K phase keeps its fragment pointers as plain integers: hipcc otherwise re-derives
// "base + index" per access with a vector add, and every vector-ALU instruction between two fp32 MFMAs idles the matrix pipe)
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ f32x4 lds_read16(int addr) { return *reinterpret_cast<lds_f32x4*>((unsigned)addr); }

constexpr int PW_WAVES = 8;
constexpr int TBUF = 16 * 128;          // wave-private transpose buffer: 16 pixel rows x 32 channels (rows of 128 B: conflict-free
                                        // for the ds_write_b32 of the accumulator layout AND the ds_read_b128 of the row layout)
constexpr unsigned kOob = 0x80000000u;

struct PwArgs {
    const float* x;        // [M, C1]
    const float* x2;       // [M, C2] or null: second source of the reduction (projection-shortcut blocks: [W_a | W_b] . [h ; x])
    const float* w;        // [K, C1 + C2] row-major, scales folded
    const float* shift;    // [K]
    const float* res;      // [M, K] (res_mode 1) | coarse map [N, rH, rW, K] (res_mode 2) | null
    float* y;              // [M, K]
    int M, C1, C2, K;
    int relu;
    int ns;                // weight slabs = K / NS
    int Ho, Wo, rH, rW;    // res_mode 2: output grid and coarse grid
    unsigned m_HoWo, m_Wo; // ceil(2^32 / d) multipliers
};

template <int MT, int NT, bool DUAL, int RES>
__global__ __launch_bounds__(64 * PW_WAVES, 1) void pw_sw_kernel(const PwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = 32 * NT;
    const int Ct = DUAL ? p.C1 + p.C2 : p.C1;
    const int LDW = Ct * 4 + 16;                 // slab row: odd number of 16-byte slots => conflict-free ds_read_b128 fragments
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int slab = (b >> 3) % p.ns;
    const int grp = (b >> 3) / p.ns;             // this block's index among the blocks of (xcd, slab)
    const int gpb = ((int)gridDim.x >> 3) / p.ns;
    const int n0 = slab * NS;

    // ---- the weight slab -> LDS, once ----------------------------------------------------------------------------------
    {
        const int vpr = Ct >> 2;
        const int total = NS * vpr;
        for (int v = tid; v < total; v += 64 * PW_WAVES) {
            const int r = v / vpr, c4 = v - r * vpr;
            const f32x4 val = *reinterpret_cast<const f32x4*>(p.w + (size_t)(n0 + r) * Ct + c4 * 4);
            *reinterpret_cast<f32x4*>(smem + r * LDW + c4 * 16) = val;
        }
    }
    __syncthreads();
    char* const tb = smem + NS * LDW + wid * TBUF;

    const int tiles = (p.M + 32 * MT - 1) / (32 * MT);
    const int tiles_x = tiles > xcd ? (tiles - xcd + 7) >> 3 : 0;      // row tiles of this XCD: xcd, xcd + 8, ...
    const int wstride = gpb * PW_WAVES;
    int tt = grp * PW_WAVES + wid;
    if (tt >= tiles_x) return;

    const int nks = Ct >> 3;                     // k-steps of 8 channels (a multiple of 4: C1, C2 are multiples of 32)
    const int nks1 = p.C1 >> 3;
    // per-lane constants
    const unsigned a_lane1 = (unsigned)((lane & 31) * p.C1 * 4 + (lane >> 5) * 16);
    const unsigned a_lane2 = (unsigned)((lane & 31) * p.C2 * 4 + (lane >> 5) * 16);
    const int b_lane = (int)(unsigned)(size_t)(lds_char*)smem + (lane & 31) * LDW + (lane >> 5) * 16;
    const int t_wr = ((lane >> 5) * 4) * 128 + (lane & 31) * 4;        // transpose write: row 4*(l>>5) (+ reg rows), col l&31
    const int t_rd = (lane >> 3) * 128 + (lane & 7) * 16;              // transpose read: row l>>3 (+8q), 4 channels at (l&7)*4
    const unsigned e_lane = (unsigned)((lane >> 3) * p.K * 4 + (lane & 7) * 16);   // y / residual: row l>>3, channels (l&7)*4
    const unsigned s_lane = (unsigned)((lane & 7) * 16);                           // shift vector: channels (l&7)*4
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.shift, 0, p.K * 4, 0x00020000);

    // A stream: (tile, k-step) pairs in order; loads run 4 k-steps ahead of the MFMAs, across tile boundaries.  The descriptor
    // of the tile being fetched covers exactly its rows (32-bit scalar arithmetic; rows past M and tiles past the end of this
    // wave's list have no records: the loads stay unconditional and return zeros nobody uses)
    f32x4 ring[4][MT];
    int ld_tt = tt, ld_ks = 0;                   // position of the NEXT load of the stream
    __amdgpu_buffer_rsrc_t ld_rs1, ld_rs2;
    auto set_ld_tile = [&](int t) {
        const int row0 = (xcd + 8 * t) * (32 * MT);
        const int rows = t < tiles_x ? min(32 * MT, p.M - row0) : 0;
        ld_rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)row0 * p.C1), 0, rows * p.C1 * 4, 0x00020000);
        if constexpr (DUAL)
            ld_rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 + (size_t)row0 * p.C2), 0, rows * p.C2 * 4, 0x00020000);
    };
    set_ld_tile(tt);
    auto issue_a = [&](f32x4 (&slot)[MT]) {
        if constexpr (DUAL) {
            const bool second = ld_ks >= nks1;           // wave-uniform: scalar selects
            const int Cs = second ? p.C2 : p.C1;
            const int ks = second ? ld_ks - nks1 : ld_ks;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                slot[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? ld_rs2 : ld_rs1, second ? a_lane2 : a_lane1,
                                                                                         ks * 32 + i * 32 * Cs * 4, 0));
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
                slot[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ld_rs1, a_lane1, ld_ks * 32 + i * 32 * p.C1 * 4, 0));
        }
        if (++ld_ks == nks) { ld_ks = 0; ld_tt += wstride; set_ld_tile(ld_tt); }
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) issue_a(ring[s]);

    f32x16 acc[MT][NT];
    f32x4 fb[NT];
    // B fragment pointers: slab row (32 j + l & 31), k-slot (l >> 5), at the k-step the current trip starts with
    int bj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bj[j] = b_lane + j * 32 * LDW;
        asm volatile("" : "+v"(bj[j]));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = lds_read16(bj[j]);

    // four k-steps (one trip around the A ring).  FIRST: the tile's first trip -- its first MFMA per accumulator takes the
    // inline constant 0 as C operand instead of 128 v_mov zeroing instructions
    auto trip = [&](int ks0, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u == 3) {
                // the fragments fetched during this k-step are the next trip's first: move the pointers on (past the last
                // k-step: back to step 0, the next tile's first) -- the K phase's only vector-ALU instructions, in one group
                const int inc = ks0 + 4 < nks ? 128 : -(nks - 4) * 32;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    bj[j] += inc;
                    asm volatile("" : "+v"(bj[j]));      // keep it ONE add per pointer per trip (hipcc otherwise re-derives every
                                                         // address from the loop counter: one v_add per ds_read)
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        if (FIRST && u == 0 && kk == 0) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][i][kk], fb[j][kk], z, 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
                        }
                    }
                // this n-tile's fragment is consumed: fetch the next k-step's into the same registers (the other NT - 1
                // n-tiles of MFMAs cover the LDS latency)
                fb[j] = lds_read16(bj[j] + (u < 3 ? (u + 1) * 32 : 0));
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_a(ring[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    for (; tt < tiles_x; tt += wstride) {
        // ---- K phase: MFMAs + LDS reads + global loads only ---------------------------------------------------------------
        trip(0, std::true_type{});
        for (int ks0 = 4; ks0 < nks; ks0 += 4) trip(ks0, std::false_type{});

        // ---- epilogue: y = act(acc + shift [+ residual]) through the wave-private transpose ------------------------------
        // Steps (i, h, j) = half an MFMA tile each: 16 pixel rows x 32 channels; a lane handles rows (l >> 3) and 8 + (l >> 3),
        // 4 channels at (l & 7) * 4.  Every load is unconditional (a branch around a load costs a vmcnt(0)); the residual and
        // shift vectors of step s + 1 are requested before step s is processed.
        const int row0 = (xcd + 8 * tt) * (32 * MT);
        const int ybytes = min(32 * MT, p.M - row0) * p.K * 4;       // rows past M: out of range (loads return 0, stores are dropped)
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)row0 * p.K), 0, ybytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((RES == 1 ? p.res : p.y) + (size_t)row0 * p.K), 0, ybytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(RES == 2 ? p.res : p.y), 0, (int)kOob, 0x00020000);
        constexpr int STEPS = MT * 2 * NT;
        unsigned uo[MT * 2][2];           // RES == 2: offsets of the coarse pixels under this lane's rows
        if constexpr (RES == 2) {
            // img0 = image of the tile's first row (one wave-uniform division per tile); HoWo >= 64 (host-checked), so a row of
            // the tile lies in that image or in the next one
            const int HoWo = p.Ho * p.Wo;
            const int img0 = row0 / HoWo;
            const float fh = (float)p.rH / (float)p.Ho, fw = (float)p.rW / (float)p.Wo;
#pragma unroll
            for (int ih = 0; ih < MT * 2; ++ih)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int rl = min(row0 - img0 * HoWo + 16 * ih + 8 * q + (lane >> 3), p.M - 1 - img0 * HoWo);
                    const int nl = rl >= HoWo ? 1 : 0;
                    const int rm = rl - nl * HoWo;
                    const int ho = (int)__umulhi((unsigned)rm, p.m_Wo);
                    const int wo = rm - ho * p.Wo;
                    const int ht = min((int)floorf((float)ho * fh), p.rH - 1);
                    const int wt = min((int)floorf((float)wo * fw), p.rW - 1);
                    uo[ih][q] = (unsigned)((((img0 + nl) * p.rH + ht) * p.rW + wt) * p.K + n0 + (lane & 7) * 4) * 4u;
                }
        }
        f32x4 rv[2][2], sh[2];
        auto request = [&](int s) {       // s = (i * 2 + h) * NT + j
            const int ih = s / NT, j = s % NT;
            const int soff = (16 * ih * p.K + n0 + 32 * j) * 4;       // wave-uniform
            sh[s & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, s_lane, (n0 + 32 * j) * 4, 0));
            if constexpr (RES == 1) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    rv[s & 1][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, e_lane, soff + q * 8 * p.K * 4, 0));
            } else if constexpr (RES == 2) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    rv[s & 1][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, uo[ih][q], j * 128, 0));
            }
        };
        // (two copies of the step sequence, ReLU / no ReLU: a per-vector branch would sit between the loads)
        auto steps = [&](auto relu_tag) {
            constexpr bool RELU = decltype(relu_tag)::value;
            request(0);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const int ih = s / NT, i = ih >> 1, h = ih & 1, j = s % NT;
                const int soff = (16 * ih * p.K + n0 + 32 * j) * 4;
                if (s + 1 < STEPS) request(s + 1);
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    *reinterpret_cast<float*>(tb + t_wr + ((r & 3) + 8 * (r >> 2)) * 128) = acc[i][j][8 * h + r];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(tb + t_rd + q * 8 * 128) + sh[s & 1];
                    if constexpr (RES != 0) v += rv[s & 1][q];
                    if constexpr (RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {      // one v_max_f32 (fmaxf's result; hipcc adds a canonicalising v_max v, v to fmaxf)
                            float o;
                            asm("v_max_f32_e32 %0, 0, %1" : "=v"(o) : "v"(v[e]));
                            v[e] = o;
                        }
                    }
                    const u32x4 vbits = __builtin_bit_cast(u32x4, v);
                    __builtin_amdgcn_raw_buffer_store_b128(vbits, y_rsrc, e_lane, soff + q * 8 * p.K * 4, 0);
                    // Round 6: a 16-byte store reads its data registers over several cycles; behind one with an SGPR offset hipcc 7.2
                    // places the next piece's arithmetic without a wait state (LLVM's hazard table assumes the hazard only exists for an
                    // immediate scalar offset -- on gfx950 it does not hold: seam_pwpc.hip round 5 stored wrong fourth channels,
                    // seam_pwh.hip round 6 NaNs).  tools/isa_store_hazard.py found `buffer_store_dwordx4 v[0:3], .., s64 offen` /
                    // `v_pk_add_f32 v[2:3], ..` back to back in eight instances of this kernel; no test ever saw a wrong value (the
                    // s_waitcnt between the two happened to cover the read), but nothing guaranteed it.  The data registers are an
                    // input of the s_nop below: nothing may overwrite them before it has issued.
                    asm volatile("s_nop 1" ::"v"(vbits));
                }
            }
        };
        if (p.relu) steps(std::true_type{});
        else steps(std::false_type{});
    }
}

inline unsigned magic_u32(unsigned long long d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }

// 0 = not served by this kernel (the caller falls back to the implicit GEMM); otherwise MT * 100 + NT
inline int pw_config(int M, int C1, int C2, int K) {
    const int Ct = C1 + C2;
    if (M <= 0 || C1 <= 0 || (C1 % 32) || C2 < 0 || (C2 % 32) || Ct > 256 || K <= 0 || (K % 64)) return 0;
    const long room = 163840 - PW_WAVES * TBUF;
    // every branch needs its slab count ns = K / (32 * NT) to divide 32: the block -> (slab, row group) decode walks ns slabs inside
    // each XCD's 32 blocks; other K (768, 8448, ...) are not served here and stay on the implicit GEMM
    if (K % 256 == 0 && 256l * (Ct * 4 + 16) <= room && 32 % (K / 256) == 0) return 108;
    if (K % 128 == 0 && 128l * (Ct * 4 + 16) <= room && 32 % (K / 128) == 0) return 204;
    if (32 % (K / 64) == 0) return 202;
    return 0;
}

}  // namespace

extern "C" {

// MT * 100 + NT of the wave tile seam_conv1x1_sw_f32 will use for [M, C1 + C2] x [K, C1 + C2]^T, or 0 when the shape is not
// served (C1 + C2 > 256, C1 / C2 not multiples of 32, K not a multiple of 64).  Independent of M (> 0): a batch never changes the
// kernel an image's pixels go through.
int seam_conv1x1_sw_config(int M, int C1, int C2, int K) { return pw_config(M, C1, C2, K); }

// y[M,K] = act( [x | x2][M, C1 + C2] . w[K, C1 + C2]^T + shift [+ residual] ), exact fp32 on the matrix cores.
//   res_mode 0: no residual; 1: residual [M,K]; 2: residual = coarse NHWC map [N, rH, rW, K] added through a nearest-neighbour
//   upsample to the [Ho, Wo] output grid (M = N * Ho * Wo; ATen's index rule, as seam_conv2d_upres_f32).
int seam_conv1x1_sw_f32(const float* x, const float* x2, const float* w, const float* shift, const float* residual, float* y,
                        int M, int C1, int C2, int K, int relu, int res_mode, int Ho, int Wo, int rH, int rW, void* stream) {
    const int cfg = pw_config(M, C1, C2, K);
    if (!cfg || !shift || (C2 > 0 && (!x2 || res_mode)) || (res_mode && !residual) || res_mode < 0 || res_mode > 2)
        return (int)hipErrorInvalidValue;
    if (res_mode == 2 && (Ho <= 0 || Wo <= 0 || rH <= 0 || rW <= 0 || M % (Ho * Wo) || Ho * Wo < 64 ||
                          (unsigned long long)Ho * Wo * Wo >= (1ull << 32)))
        return (int)hipErrorInvalidValue;
    PwArgs a;
    a.x = x; a.x2 = x2; a.w = w; a.shift = shift; a.res = residual; a.y = y;
    a.M = M; a.C1 = C1; a.C2 = C2; a.K = K; a.relu = relu;
    a.Ho = Ho; a.Wo = Wo; a.rH = rH; a.rW = rW;
    a.m_HoWo = res_mode == 2 ? magic_u32((unsigned long long)Ho * Wo) : 0;
    a.m_Wo = res_mode == 2 ? magic_u32((unsigned long long)Wo) : 0;
    const int MT = cfg / 100, NT = cfg % 100;
    a.ns = K / (32 * NT);
    const int Ct = C1 + C2;
    const size_t lds = (size_t)(32 * NT) * (Ct * 4 + 16) + PW_WAVES * TBUF;
    // one block per CU; fewer when the rows do not fill them (a block's first act is to copy its slab): blocks per slab = row
    // tiles / 8 waves, rounded up to the 8 XCDs.  The grid never changes a result (each output pixel is one wave's fixed fma chain).
    const int max_blk = seam_opt::get(seam_opt::PW_BLOCKS);      // dev knob; a multiple of 64
    if (max_blk < 64 || max_blk % 64 || max_blk < 8 * a.ns) return (int)hipErrorInvalidValue;
    const int tiles = (M + 32 * MT - 1) / (32 * MT);
    int per_slab = (((tiles + PW_WAVES - 1) / PW_WAVES + 7) / 8) * 8;
    if (per_slab > max_blk / a.ns) per_slab = (max_blk / a.ns) & ~7;      // a multiple of 8 (one row group per XCD), >= 8 by the check above
    const int nblk = per_slab * a.ns;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define SEAM_PW_LAUNCH(mt, nt, dual, res)                                                                                         \
    do {                                                                                                                         \
        static std::atomic<unsigned> attr_done{0};      /* one bit per device: the ABI is thread-safe per stream */               \
        int dev_ = 0;                                                                                                            \
        (void)hipGetDevice(&dev_);                                                                                               \
        if (!(attr_done.load(std::memory_order_acquire) & (1u << (dev_ & 31)))) {                                                \
            e = hipFuncSetAttribute((const void*)pw_sw_kernel<mt, nt, dual, res>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                    163840);                                                                                     \
            if (e == hipSuccess) attr_done.fetch_or(1u << (dev_ & 31), std::memory_order_release);                               \
        }                                                                                                                        \
        if (e == hipSuccess) hipLaunchKernelGGL((pw_sw_kernel<mt, nt, dual, res>), dim3(nblk), dim3(64 * PW_WAVES), lds, st, a); \
    } while (0)
#define SEAM_PW_CFG(dual, res)                                                                                                   \
    do {                                                                                                                         \
        if (MT == 1 && NT == 8) SEAM_PW_LAUNCH(1, 8, dual, res);                                                                 \
        else if (MT == 2 && NT == 4) SEAM_PW_LAUNCH(2, 4, dual, res);                                                            \
        else SEAM_PW_LAUNCH(2, 2, dual, res);                                                                                    \
    } while (0)
    if (C2 > 0) SEAM_PW_CFG(true, 0);
    else if (res_mode == 0) SEAM_PW_CFG(false, 0);
    else if (res_mode == 1) SEAM_PW_CFG(false, 1);
    else SEAM_PW_CFG(false, 2);
#undef SEAM_PW_CFG
#undef SEAM_PW_LAUNCH
    if (e != hipSuccess) return (int)e;
    return (int)hipGetLastError();
}

}  // extern "C"
