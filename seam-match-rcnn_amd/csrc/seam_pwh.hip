// seam_pwh.hip -- pointwise (1x1, stride 1) convolution in fp16 (fp32 accumulate) as a STREAMING kernel: weights stationary in LDS,
// activation rows in and out as 16-byte NHWC pieces, independent waves (round 6).
//
// On the config-5 (fp16) path the 1x1 layers of the ResNet bottlenecks, the FPN laterals and the heads are HBM-bound: a pixel of the
// 64 -> 256 expansion moves 128 + 512 (+ 512 residual) bytes for 32 k FLOP.  On the implicit GEMM (conv_igemm<_Float16,128,*>) they ran
// at 0.23-0.83 of the HBM roof (profiles/r05_breakdown_c5_f16.txt: 49 launches, 85 of the step's 182 ms): a 128 x 128 tile's whole K
// loop is 1-16 chunks of ~130 cycles each -- the tile is all prologue and epilogue, the four waves of a block meet at a barrier per
// chunk and leave the memory pipe idle while they finish a tile together.
//
// This is the fp16 twin of seam_pw.hip (conv1x1_sw, exact fp32), with the same three properties:
//   * a block (8 waves, one per CU) copies its SLAB of the weight matrix -- NS = 32 NT output channels x C halves, <= 135 KB -- and
//     the slab's scale / shift vectors into LDS once and never meets a barrier again;
//   * every wave owns whole pixel rows: a wave tile is 32 MT pixels x NS channels (<= 128 fp32 accumulators).  A fragments come
//     straight from global memory in MFMA layout (v_mfma_f32_32x32x16_f16: lane = pixel row, 16 bytes = 8 consecutive k) through a
//     4-deep ring that runs ACROSS tile boundaries -- the loads of the next tile are in flight while this one is finished -- and B
//     fragments from the slab with ds_read_b128;
//   * the epilogue IS the loop: per half MFMA tile (16 pixels x 32 channels) a wave-private 2 KB LDS transpose turns the accumulator
//     layout into rows, a lane finishes 8 consecutive channels of one pixel (scale / shift in fp32, residual, ReLU, one rounding to
//     fp16) and stores them as ONE 16-byte piece; residual pieces are requested three steps ahead.  Nothing waits for anything but
//     its own loads, and the other seven waves of the CU are at unrelated points of their own tiles.
// Arithmetic: the same products as seam_conv2d_f16 (fp16 operands, fp32 accumulation, fp32 scale / shift / residual, one rounding),
// accumulated in a different order: results agree to fp32 rounding of the accumulation, not bit for bit; an output is one wave's
// fixed fma chain, so results are deterministic and independent of M.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ f32x4 lds_read16(int addr) { return *reinterpret_cast<lds_f32x4*>((unsigned)addr); }

constexpr int PWH_WAVES = 8;
constexpr int TBUF = 16 * 128;          // wave-private transpose buffer: 16 pixel rows x 32 channels fp32
// residual pieces in flight (epilogue steps ahead): the layers are HBM-bound and a wave's reads in flight are its only lever on the
// memory pipe (8 waves per CU x 6-8 KB at ~2 us = ~6 TB/s chip-wide); bounded by the register file (256 per wave at 8 waves per CU)
template <int MT> struct PwhCfg { static constexpr int RESQ = MT == 1 ? 5 : MT == 2 ? 4 : 2; };

struct PwhArgs {
    const _Float16* x;     // [M, C1]
    const _Float16* x2;    // [M, C2] or null: second source of the reduction (stride-1 projection-shortcut block)
    const _Float16* w;     // [K, C1 + C2] row-major fp16
    const float* scale;    // [K] or null (= 1)
    const float* shift;    // [K] or null (= 0)
    const _Float16* res;   // [M, K] (RES 1) | coarse NHWC map [N, rH, rW, K] added through a nearest-neighbour upsample (RES 2) | null
    _Float16* y;           // [M, K]
    int M, C1, C2, K;
    int relu;
    int ns;                // weight slabs = K / NS
    int Ho, Wo, rH, rW;    // RES 2: output grid and coarse grid
    unsigned m_Wo;         // ceil(2^32 / Wo)
};

template <int MT, int NT, bool DUAL, int RES>
__global__ __launch_bounds__(64 * PWH_WAVES, 1) void pw_swh_kernel(const PwhArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = 32 * NT;
    const int Ct = DUAL ? p.C1 + p.C2 : p.C1;
    const int LDW = Ct * 2 + 16;                 // slab row: odd number of 16-byte slots => conflict-free ds_read_b128 fragments
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int slab = (b >> 3) % p.ns;
    const int grp = (b >> 3) / p.ns;             // this block's index among the blocks of (xcd, slab)
    const int gpb = ((int)gridDim.x >> 3) / p.ns;
    const int n0 = slab * NS;

    // ---- the weight slab and its epilogue vectors -> LDS, once -----------------------------------------------------------
    float* const vec = reinterpret_cast<float*>(smem + NS * LDW + PWH_WAVES * TBUF);       // [2][NS]: scale, shift
    {
        const int vpr = Ct >> 3;                 // 16-byte pieces per row
        const int total = NS * vpr;
        for (int v = tid; v < total; v += 64 * PWH_WAVES) {
            const int r = v / vpr, c8 = v - r * vpr;
            const f32x4 val = *reinterpret_cast<const f32x4*>(p.w + (size_t)(n0 + r) * Ct + c8 * 8);
            *reinterpret_cast<f32x4*>(smem + r * LDW + c8 * 16) = val;
        }
        for (int v = tid; v < NS; v += 64 * PWH_WAVES) {
            vec[v] = p.scale ? p.scale[n0 + v] : 1.f;
            vec[NS + v] = p.shift ? p.shift[n0 + v] : 0.f;
        }
    }
    __syncthreads();
    char* const tb = smem + NS * LDW + wid * TBUF;

    const int tiles = (p.M + 32 * MT - 1) / (32 * MT);
    const int tiles_x = tiles > xcd ? (tiles - xcd + 7) >> 3 : 0;      // row tiles of this XCD: xcd, xcd + 8, ...
    const int wstride = gpb * PWH_WAVES;
    int tt = grp * PWH_WAVES + wid;
    if (tt >= tiles_x) return;

    const int nks = Ct >> 4;                     // k-steps of 16 channels (a multiple of 4: C1, C2 are multiples of 64)
    const int nks1 = p.C1 >> 4;
    // per-lane constants
    const unsigned a_lane1 = (unsigned)((lane & 31) * p.C1 * 2 + (lane >> 5) * 16);
    const unsigned a_lane2 = (unsigned)((lane & 31) * p.C2 * 2 + (lane >> 5) * 16);
    const int b_lane = (int)(unsigned)(size_t)(lds_char*)smem + (lane & 31) * LDW + (lane >> 5) * 16;
    // transpose buffer [16 rows][32 channels] fp32; 16-byte slot q of row r sits at slot q ^ (r & 1): a lane's two ds_read_b128
    // (row l >> 2, slots 2 (l & 3) and 2 (l & 3) + 1) are conflict-free -- eight consecutive lanes = two rows = every bank once --
    // and so are the ds_write_b32 of the accumulator layout (one row, a permutation of its 32 columns, per 32 lanes)
    const int t_wr0 = ((lane >> 5) * 4) * 128 + ((lane & 31) >> 2) * 16 + (lane & 3) * 4;            // even rows: row 4*(l>>5) (+ reg rows), col l&31
    const int t_wr1 = ((lane >> 5) * 4) * 128 + ((((lane & 31) >> 2)) ^ 1) * 16 + (lane & 3) * 4;    // odd rows
    const int t_par = (lane >> 2) & 1;
    const int t_rd0 = (lane >> 2) * 128 + (((lane & 3) * 2) ^ t_par) * 16;          // transpose read: row l>>2, channels (l&3)*8 .. +3
    const int t_rd1 = (lane >> 2) * 128 + (((lane & 3) * 2 + 1) ^ t_par) * 16;      // ... +4 .. +7
    const unsigned e_lane = (unsigned)((lane >> 2) * p.K * 2 + (lane & 3) * 16);   // y / residual: row l>>2, channels (l&3)*8
    const float* const v_lane = vec + (lane & 3) * 8;

    // A stream: (tile, k-step) pairs in order; loads run 4 k-steps ahead of the MFMAs, across tile boundaries.  The descriptor
    // of the tile being fetched covers exactly its rows (rows past M and tiles past the end of this wave's list have no records:
    // the loads stay unconditional and return zeros nobody uses)
    constexpr int RD = 4;        // ring depth in k-steps (8 for the one-row-tile form was tried: hipcc spills ~56 registers per lane)
    f16x8 ring[RD][MT];
    int ld_tt = tt, ld_ks = 0;                   // position of the NEXT load of the stream
    // (gfx950 range-checks vector offset + scalar offset + immediate against the descriptor's size -- tools/probes/soffset_probe.hip,
    //  profiles/r06_soffset_probe.txt -- so the row-group step may ride in the scalar offset: rows past M read as zeros)
    __amdgpu_buffer_rsrc_t ld_rs1, ld_rs2;
    auto set_ld_tile = [&](int t) {
        const int row0 = (xcd + 8 * t) * (32 * MT);
        const int rows = t < tiles_x ? min(32 * MT, p.M - row0) : 0;
        ld_rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)row0 * p.C1), 0, rows * p.C1 * 2, 0x00020000);
        if constexpr (DUAL)
            ld_rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 + (size_t)row0 * p.C2), 0, rows * p.C2 * 2, 0x00020000);
    };
    set_ld_tile(tt);
    auto issue_a = [&](f16x8 (&slot)[MT]) {
        if constexpr (DUAL) {
            const bool second = ld_ks >= nks1;           // wave-uniform: scalar selects
            const int Cs = second ? p.C2 : p.C1;
            const int ks = second ? ld_ks - nks1 : ld_ks;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                slot[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(second ? ld_rs2 : ld_rs1, second ? a_lane2 : a_lane1,
                                                                                         ks * 32 + i * 32 * Cs * 2, 0));
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
                slot[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(ld_rs1, a_lane1, ld_ks * 32 + i * 32 * p.C1 * 2, 0));
        }
        if (++ld_ks == nks) { ld_ks = 0; ld_tt += wstride; set_ld_tile(ld_tt); }
    };
#pragma unroll
    for (int s = 0; s < RD; ++s) issue_a(ring[s]);

    f32x16 acc[MT][NT];
    f16x8 fb[NT];
    int bj0 = b_lane;                            // B fragment pointer of n-tile 0: slab row (l & 31), k-slot (l >> 5), current trip; n-tile j
    const int bstep = 32 * LDW;                  // sits bstep * j further (one v_add per read: nothing here is bound by the vector ALU)

    // four k-steps (one trip around the A ring).  FIRST: the tile's first trip multiplies into the constant 0.  LAST: the tile's last
    // trip does not fetch B fragments past its end -- the next tile reads its first ones itself, so that none are carried (32
    // registers) across the epilogue, where the residual ring needs them.
    auto trip = [&](int ks0, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const bool last = ks0 + 4 >= nks;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u == 3) {
                const int inc = !last ? 128 : -(nks - 4) * 32;      // past the last k-step: back to step 0 (the next tile's first)
                bj0 += inc;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const f16x8 av = ring[u][i];
                    if (FIRST && u == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, fb[j], z, 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, fb[j], acc[i][j], 0, 0, 0);
                    }
                }
                if (!(last && u == 3)) fb[j] = __builtin_bit_cast(f16x8, lds_read16(bj0 + j * bstep + (u < 3 ? (u + 1) * 32 : 0)));
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_a(ring[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    for (; tt < tiles_x; tt += wstride) {
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = __builtin_bit_cast(f16x8, lds_read16(bj0 + j * bstep));
        trip(0, std::true_type{});
        for (int ks0 = 4; ks0 < nks; ks0 += 4) trip(ks0, std::false_type{});

        // ---- epilogue: y = act(acc * scale + shift [+ residual]) -> fp16, through the wave-private transpose --------------
        // Steps s = (i, h, j) = half an MFMA tile each: 16 pixel rows x 32 channels; a lane finishes row (l >> 2), 8 channels at
        // (l & 3) * 8: one 16-byte store.  Loads are unconditional; the residual piece of step s + RESQ is requested before step s.
        const int row0 = (xcd + 8 * tt) * (32 * MT);
        const int ybytes = min(32 * MT, p.M - row0) * p.K * 2;       // rows past M: out of range (loads return 0, stores are dropped)
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)row0 * p.K), 0, ybytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((RES == 1 ? p.res : p.y) + (size_t)row0 * p.K), 0, ybytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(RES == 2 ? p.res : p.y), 0, (int)0x80000000u, 0x00020000);
        unsigned uo[MT * 2];              // RES == 2: byte offset of the coarse pixel under this lane's row of each half tile
        if constexpr (RES == 2) {
            // FPN top-down merge [TV FeaturePyramidNetwork.forward: inner + F.interpolate(top, size, "nearest")]: ATen's index rule
            // src = min(floor(dst * in / out), in - 1), as seam_conv2d_upres_f32.  img0 = image of the tile's first row (one wave-
            // uniform division per tile); Ho * Wo >= 32 * MT (host-checked): a row of the tile lies in that image or in the next one
            const int HoWo = p.Ho * p.Wo;
            const int img0 = row0 / HoWo;
            const float fh = (float)p.rH / (float)p.Ho, fw = (float)p.rW / (float)p.Wo;
#pragma unroll
            for (int ih = 0; ih < MT * 2; ++ih) {
                const int rl = min(row0 - img0 * HoWo + 16 * ih + (lane >> 2), p.M - 1 - img0 * HoWo);
                const int nl = rl >= HoWo ? 1 : 0;
                const int rm = rl - nl * HoWo;
                const int ho = (int)__umulhi((unsigned)rm, p.m_Wo);
                const int wo = rm - ho * p.Wo;
                const int ht = min((int)floorf((float)ho * fh), p.rH - 1);
                const int wt = min((int)floorf((float)wo * fw), p.rW - 1);
                uo[ih] = (unsigned)((((img0 + nl) * p.rH + ht) * p.rW + wt) * p.K + n0 + (lane & 3) * 8) * 2u;
            }
        }
        constexpr int STEPS = MT * 2 * NT;
        auto soff_of = [&](int s) { return (16 * (s / NT) * p.K + n0 + 32 * (s % NT)) * 2; };      // wave-uniform byte offset of a step
        constexpr int RESQ = PwhCfg<MT>::RESQ;
        u32x4 rv[RESQ];
        auto request = [&](int s) {
            if constexpr (RES == 1) rv[s % RESQ] = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, e_lane, soff_of(s), 0);
            if constexpr (RES == 2) rv[s % RESQ] = __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, uo[s / NT], (s % NT) * 64, 0);
        };
        auto steps = [&](auto relu_tag) {
            constexpr bool RELU = decltype(relu_tag)::value;
#pragma unroll
            for (int s = 0; s < RESQ && s < STEPS; ++s) request(s);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const int ih = s / NT, i = ih >> 1, h = ih & 1, j = s % NT;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    *reinterpret_cast<float*>(tb + ((r & 1) ? t_wr1 : t_wr0) + ((r & 3) + 8 * (r >> 2)) * 128) = acc[i][j][8 * h + r];
                const f32x4 sc0 = *reinterpret_cast<const f32x4*>(v_lane + 32 * j), sc1 = *reinterpret_cast<const f32x4*>(v_lane + 32 * j + 4);
                const f32x4 sh0 = *reinterpret_cast<const f32x4*>(v_lane + NS + 32 * j), sh1 = *reinterpret_cast<const f32x4*>(v_lane + NS + 32 * j + 4);
                f32x4 v0 = *reinterpret_cast<const f32x4*>(tb + t_rd0) * sc0 + sh0;
                f32x4 v1 = *reinterpret_cast<const f32x4*>(tb + t_rd1) * sc1 + sh1;
                if constexpr (RES != 0) {
                    const f16x8 rh = __builtin_bit_cast(f16x8, rv[s % RESQ]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] += (float)rh[e]; v1[e] += (float)rh[e + 4]; }
                    if (s + RESQ < STEPS) request(s + RESQ);
                }
                if constexpr (RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                }
                f16x8 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { hv[e] = (_Float16)v0[e]; hv[e + 4] = (_Float16)v1[e]; }
                // (the step offset rides in the VECTOR offset: behind a 16-byte store with an SGPR offset hipcc puts the next step's
                //  arithmetic without the wait state the store's data registers need -- LLVM's hazard table only covers an immediate
                //  scalar offset -- and the store then writes the NEXT step's values: seen here as NaNs in the <2,4> residual form, in
                //  seam_pwpc.hip as wrong fourth channels.  tools/isa_store_hazard.py scans the library's ISA for the pattern.)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), y_rsrc, e_lane + (unsigned)soff_of(s), 0, 0);
            }
        };
        if (p.relu) steps(std::true_type{});
        else steps(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The ResNet stem on the space-to-depth frame as the same streaming kernel (round 6).
// ResNet.conv1 (7x7 / stride 2 / pad 3) on the frame seam_preprocess_s2d_batch_f16 writes is a 4 x 4 / stride-1 convolution over
// 16-channel cells (seam_conv2d_crop_f16): 16 taps x 16 channels = 16 k-steps of v_mfma_f32_32x32x16_f16, 64 output channels.  On a
// frame PADDED with zero cells (2 before, 1 after, in both directions; pitch Wp cells) the tap (r, s) of output position m (flattened
// over the padded grid, m = (n Hp + y) Wp + x) is the cell m + r Wp + s: a tap is a constant shift of the flattened index, so a
// wave tile = 32 MT CONSECUTIVE positions reads, per tap, one contiguous run of 32 MT cells (1 KB per load instruction, 16 bytes per
// lane) -- the implicit GEMM gathered the sixteen 32-byte cells of every output pixel one by one and ran at 0.23 of the HBM roof.
// Positions in the padding columns / rows (1.2 % of the grid) are computed and not stored; the output is the dense NHWC map.
struct StemArgs {
    const _Float16* x;     // padded frame [N, Hp, Wp, 16]
    const _Float16* w;     // [64, 256] row-major fp16, k = (4 r + s) * 16 + channel
    const float* scale;    // [64] or null
    const float* shift;    // [64] or null
    _Float16* y;           // [N, Ho, Wo, 64]
    int N, Hp, Wp, Ho, Wo, relu;
    long long Mp;          // N * Hp * Wp
    unsigned m_HpWp, m_Wp; // ceil(2^32 / d)
};

__global__ __launch_bounds__(64 * PWH_WAVES, 1) void stem_swh_kernel(const StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = 4, NT = 2, NS = 64, Ct = 256, LDW = Ct * 2 + 16, RD = 4, nks = 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const vec = reinterpret_cast<float*>(smem + NS * LDW + PWH_WAVES * TBUF);       // [2][NS]: scale, shift
    for (int v = tid; v < NS * (Ct >> 3); v += 64 * PWH_WAVES) {
        const int r = v / (Ct >> 3), c8 = v - r * (Ct >> 3);
        *reinterpret_cast<f32x4*>(smem + r * LDW + c8 * 16) = *reinterpret_cast<const f32x4*>(p.w + (size_t)r * Ct + c8 * 8);
    }
    for (int v = tid; v < NS; v += 64 * PWH_WAVES) {
        vec[v] = p.scale ? p.scale[v] : 1.f;
        vec[NS + v] = p.shift ? p.shift[v] : 0.f;
    }
    __syncthreads();
    char* const tb = smem + NS * LDW + wid * TBUF;

    // tiles of 128 consecutive flattened positions, dealt to the block's waves in block-contiguous order
    const long long tiles = (p.Mp + 32 * MT - 1) / (32 * MT);
    const long long wstride = (long long)gridDim.x * PWH_WAVES;
    long long tt = (long long)blockIdx.x * PWH_WAVES + wid;
    if (tt >= tiles) return;

    const unsigned a_lane = (unsigned)((lane & 31) * 32 + (lane >> 5) * 16);
    const int b_lane = (int)(unsigned)(size_t)(lds_char*)smem + (lane & 31) * LDW + (lane >> 5) * 16;
    const int t_wr0 = ((lane >> 5) * 4) * 128 + ((lane & 31) >> 2) * 16 + (lane & 3) * 4;
    const int t_wr1 = ((lane >> 5) * 4) * 128 + ((((lane & 31) >> 2)) ^ 1) * 16 + (lane & 3) * 4;
    const int t_par = (lane >> 2) & 1;
    const int t_rd0 = (lane >> 2) * 128 + (((lane & 3) * 2) ^ t_par) * 16;
    const int t_rd1 = (lane >> 2) * 128 + (((lane & 3) * 2 + 1) ^ t_par) * 16;
    const float* const v_lane = vec + (lane & 3) * 8;

    f16x8 ring[RD][MT];
    long long ld_tt = tt;
    int ld_ks = 0;
    __amdgpu_buffer_rsrc_t ld_rs;
    auto set_ld_tile = [&](long long t) {            // everything from the tile's first position to the end of the frame (<= 2^31 - 1 bytes)
        const long long row0 = t * (32 * MT);
        const long long left = t < tiles ? (p.Mp - row0) * 32 : 0;
        ld_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)(t < tiles ? row0 : 0) * 16), 0,
                                                  (int)(left > 0x7fffffffLL ? 0x7fffffffLL : left), 0x00020000);
    };
    set_ld_tile(tt);
    auto issue_a = [&](f16x8 (&slot)[MT]) {          // k-step ld_ks = tap (r, s) = (ld_ks >> 2, ld_ks & 3): the cells r Wp + s further on
        const int so = ((ld_ks >> 2) * p.Wp + (ld_ks & 3)) * 32;
#pragma unroll
        for (int i = 0; i < MT; ++i)
            slot[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(ld_rs, a_lane, so + i * 32 * 32, 0));
        if (++ld_ks == nks) { ld_ks = 0; ld_tt += wstride; set_ld_tile(ld_tt); }
    };
#pragma unroll
    for (int s = 0; s < RD; ++s) issue_a(ring[s]);

    f32x16 acc[MT][NT];
    f16x8 fb[NT];
    int bj0 = b_lane;
    constexpr int bstep = 32 * LDW;
    auto trip = [&](int ks0, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const bool last = ks0 + 4 >= nks;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u == 3) bj0 += !last ? 128 : -(nks - 4) * 32;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (FIRST && u == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[u][i], fb[j], z, 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[u][i], fb[j], acc[i][j], 0, 0, 0);
                    }
                }
                if (!(last && u == 3)) fb[j] = __builtin_bit_cast(f16x8, lds_read16(bj0 + j * bstep + (u < 3 ? (u + 1) * 32 : 0)));
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_a(ring[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int HpWp = p.Hp * p.Wp;
    for (; tt < tiles; tt += wstride) {
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = __builtin_bit_cast(f16x8, lds_read16(bj0 + j * bstep));
        trip(0, std::true_type{});
        for (int ks0 = 4; ks0 < nks; ks0 += 4) trip(ks0, std::false_type{});

        // ---- epilogue: the dense position of every row of the tile (or none: padding columns / rows), then as conv1x1_swh ----
        const long long row0 = tt * (32 * MT);
        const int n0 = (int)(row0 / HpWp);                                  // image of the tile's first position (wave-uniform)
        const int rem0 = (int)(row0 - (long long)n0 * HpWp);
        const int y0 = rem0 / p.Wp;
        // dense base of the tile: the first pixel of output row min(y0, Ho - 1) of image n0; every valid position of the tile lies at
        // or behind it (a tile is 128 positions: at most into the next image), within a few rows -- 32-bit lane offsets from there
        const long long base_d = ((long long)n0 * p.Ho + min(y0, p.Ho - 1)) * p.Wo;
        const long long left_d = ((long long)p.N * p.Ho * p.Wo - base_d) * 128;
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.y + (size_t)base_d * 64), 0, (int)(left_d > 0x7fffffffLL ? 0x7fffffffLL : left_d), 0x00020000);
        unsigned uo[MT * 2];
#pragma unroll
        for (int ih = 0; ih < MT * 2; ++ih) {
            const int rl = rem0 + 16 * ih + (lane >> 2);                    // position relative to image n0's padded grid (may run into n0 + 1)
            const int nl = rl >= HpWp ? 1 : 0;
            const int rm = rl - nl * HpWp;
            const int yy = (int)__umulhi((unsigned)rm, p.m_Wp);
            const int xx = rm - yy * p.Wp;
            const bool ok = yy < p.Ho && xx < p.Wo && n0 + nl < p.N;
            const long long d = ((long long)(n0 + nl) * p.Ho + yy) * p.Wo + xx - base_d;
            uo[ih] = ok ? (unsigned)(d * 128) + (unsigned)((lane & 3) * 16) : 0x80000000u;
        }
        constexpr int STEPS = MT * 2 * NT;
        auto steps = [&](auto relu_tag) {
            constexpr bool RELU = decltype(relu_tag)::value;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const int ih = s / NT, i = ih >> 1, h = ih & 1, j = s % NT;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    *reinterpret_cast<float*>(tb + ((r & 1) ? t_wr1 : t_wr0) + ((r & 3) + 8 * (r >> 2)) * 128) = acc[i][j][8 * h + r];
                const f32x4 sc0 = *reinterpret_cast<const f32x4*>(v_lane + 32 * j), sc1 = *reinterpret_cast<const f32x4*>(v_lane + 32 * j + 4);
                const f32x4 sh0 = *reinterpret_cast<const f32x4*>(v_lane + NS + 32 * j), sh1 = *reinterpret_cast<const f32x4*>(v_lane + NS + 32 * j + 4);
                f32x4 v0 = *reinterpret_cast<const f32x4*>(tb + t_rd0) * sc0 + sh0;
                f32x4 v1 = *reinterpret_cast<const f32x4*>(tb + t_rd1) * sc1 + sh1;
                if constexpr (RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                }
                f16x8 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { hv[e] = (_Float16)v0[e]; hv[e + 4] = (_Float16)v1[e]; }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), y_rsrc, uo[ih] + (unsigned)(j * 64), 0, 0);
            }
        };
        if (p.relu) steps(std::true_type{});
        else steps(std::false_type{});
    }
}

// (Round 6, null result: the same flattened-shift form for the 3x3 / 64 -> 64 layers of layer1 on the dense map -- a tap is a constant
// shift of the flattened pixel index, windows that leave the image get an out-of-range offset -- was built, bit-identical to the
// implicit GEMM, and 0.81x its speed (profiles/r06_shift3_ab.txt: 3 015 vs 2 443 us at 240 x 192 x 336): with 128-byte cells a wave's
// A-fragment load touches 32 cache lines for 32 bytes each (the stem's 32-byte cells: one contiguous KB), four times the work in the
// texture addresser.  Removed; those layers need their patch staged through LDS like conv3x3_f16pc.)

// 0 = not served by this kernel (the caller stays on the implicit GEMM); otherwise MT * 100 + NT
inline int pwh_config(long long M, int C1, int C2, int K) {
    const int Ct = C1 + C2;
    // Ct <= 512: with the 64-channel slab a longer reduction needs (Ct = 1024) every A fragment feeds two MFMAs only and each input row
    // is re-read by K / 64 slab groups -- the kernel is then bound by the L2, and measured 0.65x the implicit GEMM
    // (profiles/r06_pwh_ab.txt: 1024 -> 256 at 48 x 84: 1274 vs 824 us).  Those layers stay on conv_igemm<_Float16,128,128>.
    if (M <= 0 || M > 0x7fffffffLL / 64 || C1 <= 0 || (C1 % 64) || C2 < 0 || (C2 % 64) || Ct > 512 || K <= 0 || (K % 64)) return 0;
    const long room = 163840 - PWH_WAVES * TBUF;
    auto fits = [&](int ns) { return (long)ns * (Ct * 2 + 16) + 2L * ns * 4 <= room; };
    // (every branch needs its slab count K / NS to divide an XCD's 32 blocks: the block -> (slab, row group) decode walks the slabs
    //  inside each XCD's blocks)
    if (K % 256 == 0 && fits(256) && 32 % (K / 256) == 0) return 108;
    if (K % 128 == 0 && fits(128) && 32 % (K / 128) == 0) return 204;
    // 64-channel slabs (K not a multiple of 128): short reductions only -- 256 -> 64 measured 0.90x the implicit GEMM, 64 -> 64 1.21x
    if (Ct <= 128 && fits(64) && 32 % (K / 64) == 0) return 402;
    return 0;
}

}  // namespace

extern "C" {

// MT * 100 + NT of the wave tile seam_conv1x1_swh_f16 will use for [M, C1 + C2] x [K, C1 + C2]^T, or 0 when the shape is not served
// (C1 / C2 not multiples of 64, C1 + C2 > 512 -- > 128 when K is not a multiple of 128 --, K not a multiple of 64 or a slab count that
// does not divide 32).  Independent of M.
int seam_conv1x1_swh_config(long long M, int C1, int C2, int K) { return pwh_config(M, C1, C2, K); }

int seam_conv1x1_swh_f16(const void* x, const void* x2, const void* w, const float* scale, const float* shift, const void* residual,
                         void* y, long long M, int C1, int C2, int K, int relu, int res_mode, int Ho, int Wo, int rH, int rW,
                         void* stream) {
    const int cfg = pwh_config(M, C1, C2, K);
    if (!cfg || (C2 > 0 && !x2) || relu < 0 || relu > 1 || res_mode < 0 || res_mode > 2 || (res_mode != 0) != (residual != nullptr))
        return (int)hipErrorInvalidValue;
    if (res_mode == 2 && (Ho <= 0 || Wo <= 0 || rH <= 0 || rW <= 0 || M % ((long long)Ho * Wo) || (long long)Ho * Wo < 128 ||
                          (unsigned long long)Ho * Wo * Wo >= (1ull << 32) || (M / ((long long)Ho * Wo)) * rH * rW * K * 2 >= (1ll << 31)))
        return (int)hipErrorInvalidValue;
    PwhArgs a;
    a.x = (const _Float16*)x; a.x2 = (const _Float16*)x2; a.w = (const _Float16*)w; a.scale = scale; a.shift = shift;
    a.res = (const _Float16*)residual; a.y = (_Float16*)y;
    a.M = (int)M; a.C1 = C1; a.C2 = C2; a.K = K; a.relu = relu;
    a.Ho = Ho; a.Wo = Wo; a.rH = rH; a.rW = rW;
    a.m_Wo = res_mode == 2 && Wo > 1 ? (unsigned)(((1ull << 32) + (unsigned)Wo - 1) / (unsigned)Wo) : 0;
    const int MT = cfg / 100, NT = cfg % 100;
    a.ns = K / (32 * NT);
    const int Ct = C1 + C2;
    const size_t lds = (size_t)(32 * NT) * (Ct * 2 + 16) + PWH_WAVES * TBUF + 2 * (32 * NT) * 4;
    const int max_blk = 256;
    if (max_blk < 8 * a.ns) return (int)hipErrorInvalidValue;
    const int tiles = (int)((M + 32 * MT - 1) / (32 * MT));
    int per_slab = (((tiles + PWH_WAVES - 1) / PWH_WAVES + 7) / 8) * 8;
    if (per_slab > max_blk / a.ns) per_slab = (max_blk / a.ns) & ~7;      // a multiple of 8 (one row group per XCD), >= 8
    const int nblk = per_slab * a.ns;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define SEAM_PWH_LAUNCH(mt, nt, dual, res)                                                                                        \
    do {                                                                                                                         \
        static std::atomic<unsigned> attr_done{0};      /* one bit per device: the ABI is thread-safe per stream */               \
        int dev_ = 0;                                                                                                            \
        (void)hipGetDevice(&dev_);                                                                                               \
        if (!(attr_done.load(std::memory_order_acquire) & (1u << (dev_ & 31)))) {                                                \
            e = hipFuncSetAttribute((const void*)pw_swh_kernel<mt, nt, dual, res>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                    163840);                                                                                     \
            if (e == hipSuccess) attr_done.fetch_or(1u << (dev_ & 31), std::memory_order_release);                               \
        }                                                                                                                        \
        if (e == hipSuccess) hipLaunchKernelGGL((pw_swh_kernel<mt, nt, dual, res>), dim3(nblk), dim3(64 * PWH_WAVES), lds, st, a); \
    } while (0)
#define SEAM_PWH_CFG(dual, res)                                                                                                   \
    do {                                                                                                                         \
        if (MT == 1 && NT == 8) SEAM_PWH_LAUNCH(1, 8, dual, res);                                                                \
        else if (MT == 2 && NT == 4) SEAM_PWH_LAUNCH(2, 4, dual, res);                                                           \
        else SEAM_PWH_LAUNCH(4, 2, dual, res);                                                                                   \
    } while (0)
    if (C2 > 0) {
        if (res_mode == 2) return (int)hipErrorInvalidValue;
        if (res_mode == 1) SEAM_PWH_CFG(true, 1); else SEAM_PWH_CFG(true, 0);
    } else {
        if (res_mode == 2) SEAM_PWH_CFG(false, 2);
        else if (res_mode == 1) SEAM_PWH_CFG(false, 1);
        else SEAM_PWH_CFG(false, 0);
    }
#undef SEAM_PWH_CFG
#undef SEAM_PWH_LAUNCH
    if (e != hipSuccess) return (int)e;
    return (int)hipGetLastError();
}

// ResNet.conv1 + bn1 + relu [TV; behind models/video_matchrcnn.py:337] on the PADDED space-to-depth frame (fp16): xpad [N, Ho + 3,
// Wo + 3, 16] = the frame of seam_preprocess_s2d_batch_f16 with 2 zero cells before and 1 after in both directions; w fp16 [64, 256],
// k = (4 r + s) * 16 + channel (the re-indexed 7x7 weights of seam_conv2d_crop_f16's call site); y fp16 [N, Ho, Wo, 64] dense.
// Same products as seam_conv2d_crop_f16 on the unpadded frame, fp32 accumulation in tap-major order; deterministic.
int seam_stem_s2d_swh_f16(const void* xpad, const void* w, const float* scale, const float* shift, void* y, int N, int Ho, int Wo,
                          int relu, void* stream) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || relu < 0 || relu > 1) return (int)hipErrorInvalidValue;
    StemArgs a;
    a.x = (const _Float16*)xpad; a.w = (const _Float16*)w; a.scale = scale; a.shift = shift; a.y = (_Float16*)y;
    a.N = N; a.Ho = Ho; a.Wo = Wo; a.Hp = Ho + 3; a.Wp = Wo + 3; a.relu = relu;
    a.Mp = (long long)N * a.Hp * a.Wp;
    // (a 128-position tile must lie in one image or run into the next one only: Hp * Wp >= 128)
    if ((long long)a.Hp * a.Wp < 128 || (long long)a.Hp * a.Wp >= (1ll << 31) / 2 || (unsigned long long)a.Hp * a.Wp * a.Wp >= (1ull << 32))
        return (int)hipErrorInvalidValue;
    a.m_HpWp = 0;
    a.m_Wp = (unsigned)(((1ull << 32) + (unsigned)a.Wp - 1) / (unsigned)a.Wp);
    const size_t lds = (size_t)64 * (256 * 2 + 16) + PWH_WAVES * TBUF + 2 * 64 * 4;
    const long long tiles = (a.Mp + 127) / 128;
    long long nblk = (tiles + PWH_WAVES - 1) / PWH_WAVES;
    if (nblk > 256) nblk = 256;
    static std::atomic<unsigned> attr_done{0};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!(attr_done.load(std::memory_order_acquire) & (1u << (dev_ & 31)))) {
        const hipError_t e = hipFuncSetAttribute((const void*)stem_swh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return (int)e;
        attr_done.fetch_or(1u << (dev_ & 31), std::memory_order_release);
    }
    hipLaunchKernelGGL(stem_swh_kernel, dim3((unsigned)nblk), dim3(64 * PWH_WAVES), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // extern "C"
