// seam_pairmf.hip -- configs[2]/[3]: street-sequence descriptors [Q,256] ranked against a LARGE product bank [G,256]
// (G >= 8192), top-k only, on the fp32 matrix cores, results bit-identical to seam_pair_logits_f32 + seam_rank_topk_f32.
//
// The reference scores a pair with the pairwise classifier on squared differences (ref models/match_head.py:161-162) and ranks
// by softmax(x5)[...,1] (ref evaluate_movingfashion.py:97-99,265-269).  Only the logit DIFFERENCE orders the products:
//     d[i,j] = x5[i,j,1] - x5[i,j,0] = sum_k wd_k (a_ik - b_jk)^2 + bd,      wd = W[1] - W[0],  bd = bias[1] - bias[0]
//            = A_i + B_j + sum_k (-2 wd_k a_ik) b_jk,      A_i = sum_k wd_k a_ik^2 + bd,   B_j = sum_k wd_k b_jk^2
// i.e. one [Q,256] x [256,G] GEMM plus two rank-1 terms.  The expanded form is NOT what the reference evaluates (cancellation),
// so it is used only to FIND candidates; the winners are then re-scored with the direct form, in the exact operation order of
// seam_pair_logits_f32, and a per-query error bound proves that no product outside the candidate set can reach the top k
// (otherwise that query alone is redone with the direct form over the whole bank -- never observed on continuous data).
//
//   1. pairmf_prep        awm2[q] = -2 wd o a_q, A_q, and P_q = sum_k (|W0_k| + |W1_k|) a_qk^2 (for the error bound)
//   2. pairmf_kernel<1>   d' of all queries against 4096 bank rows sampled in 16-row groups across the bank -> dense [Q,4096]
//   3. pairmf_thresh      tau_q = kk-th largest sampled d' (kk = k + margin): a lower bound of the final kk-th largest (two-level
//                         radix select: every wave keeps the kk largest of its quarter, wave 0 selects among the 4 kk survivors)
//   4. pairmf_kernel<0>   the whole bank: every (q, j) with d' >= tau_q is a candidate (about G * kk / 4096 per query; nothing
//                         else reaches HBM).  A wave parks its finds in LDS and writes them, at the end, into the slots of
//                         (query, this block) -- no global atomics: 250 blocks appending to 256 per-query lists through
//                         same-address L2 atomics cost 16 us of serialisation (measured); only a (query, block) pair whose
//                         slots are full falls back to the atomic per-query overflow lists.
//   5. pairmf_final       per query: gathers its candidates from its slots of every block (+ overflow list) into LDS, takes
//                         the kk best by d', re-scores them exactly, orders them, writes the top k; the bound check.
//
// pairmf_kernel -- the GEMM -- is built for CDNA4's exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 32 cycles per issue):
//   * 512 threads = 8 waves (2 per SIMD, both MFMA-bound, so they alternate on the matrix pipe); each wave OWNS 32 queries
//     whose B-operand fragments (2 column tiles x 64 k-steps = 128 VGPRs) stay in registers for the whole kernel: the query
//     side is read once per block, from L2.
//   * each block walks a contiguous range of bank rows in 32-row tiles through double-buffered LDS (rows padded to 1040 B:
//     a lane's 16-byte fragment read of row i at k-slot u lands in 16-byte bank group (i + u) mod 16 -> conflict-free
//     ds_read_b128, one per 8 MFMAs); the bank streams from HBM exactly once (2 B / clk / CU -- far from any limit).
//   * k is consumed in the order k = 64 h + 4 u + e (h = lane >> 4 the MFMA's k sub-index, u the 16-byte slot, e the element),
//     the same permutation on both operands; a tile may end in a 16-row group, so a bank of G rows costs ceil(G/256/16) row
//     groups per CU -- 80 rows for G = 20 000, 2.4 % padding.  (The fp32 MFMA shares the SIMD's FMA lanes with the VALU: 16 MFMAs
//     per ds_read_b128 pair keeps the non-MFMA share of the loop near 6 %.)
//   * epilogue on the VALU beside the other wave's MFMAs: d' = acc + A_q + B_j, compare with tau_q, ballot + prefix count into LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "seam_topk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

using namespace seam_topk;

constexpr int D = 256;            // descriptor width
constexpr int QG = 256;           // queries per block: 8 waves x 32
constexpr int TR = 32;            // bank rows per LDS tile (two 16-row MFMA groups)
constexpr int LROW = D + 4;       // padded LDS row in floats (1040 B)
constexpr int NS_ROWS = 4096;     // sampled bank rows behind the thresholds
constexpr int WCAP = 512;         // per-wave candidate buffer in LDS (a tile that would not fit is appended directly)
constexpr int LCAP = 4096;        // candidates of one query that pairmf_final can hold in LDS
constexpr int RB = 32;            // candidates re-scored per batch in pairmf_final
constexpr int RLD = D + 4;        // LDS row stride of the re-scoring batch (16-byte aligned rows, 2-way conflicts at most)
constexpr int PAIRMF_MAX_K = 64;                   // seam_pair_topk_mfma_max_k()
constexpr int PAIRMF_KK_MAX = ((PAIRMF_MAX_K + 12 + 7) / 8) * 8;      // pairmf_kk(PAIRMF_MAX_K) = 80 candidates per query at most
// gamma_n = n u / (1 - n u) of the n ~ 261-rounding fp32 chains below, with ~2x headroom (512 u instead of ~261 u: the candidate lists
// barely change, eps stays far below the kk - k margin).  Rounding model assumed: every product and add of v_mfma_f32_16x16x4_f32 and of
// the VALU chains rounds to nearest with |delta| <= u = 2^-24, denormal results kept (gfx950 default mode for MFMA and for this file).
// ABS_EPS covers what a purely relative bound does not: terms whose products underflow (d * d or w * b below FLT_MIN lose up to one
// denormal ulp each, 2 x 256 terms).
constexpr float GAMMA = 512.0f * 5.9604645e-8f;
constexpr float ABS_EPS = 1024.0f * 1.17549435e-38f;

__device__ __forceinline__ float key_to_float(unsigned key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------------------- 1. prep
__global__ __launch_bounds__(256) void pairmf_prep(const float* __restrict__ a, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ awm2,
                                                   float* __restrict__ qA, float* __restrict__ qP, unsigned* __restrict__ cnt,
                                                   unsigned* __restrict__ rmax, int* __restrict__ stats, int Q) {
    __shared__ float red[4];
    const int q = blockIdx.x, k = threadIdx.x;
    const float w0 = w[k], w1 = w[D + k];
    const float wd = w1 - w0, aw = fabsf(w0) + fabsf(w1);
    const float av = q < Q ? a[(size_t)q * D + k] : 0.f;
    // fragment order: [wave group q / 32][column tile n][k-slot u][lane = 16 h + j][e] with k = 64 h + 4 u + e, so that each of a
    // wave's 32 fragment loads in pairmf_kernel is one contiguous 1 KiB read
    {
        const int wq = q >> 5, n = (q >> 4) & 1, jj = q & 15, hh = k >> 6, u = (k >> 2) & 15, e = k & 3;
        awm2[((((size_t)(wq * 2 + n) * 16 + u) * 64 + hh * 16 + jj) << 2) + e] = -2.f * wd * av;
    }
    const float A = block_sum256(wd * av * av, red);
    const float P = block_sum256(aw * av * av, red);
    if (k == 0) {
        qA[q] = A + (bias[1] - bias[0]);
        qP[q] = P;
        cnt[q] = 0u;
        if (q == 0) {
            rmax[0] = 0u;
            if (stats) { stats[0] = 0; stats[1] = 0; stats[2] = 0; stats[3] = 0; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- 2/4. the GEMM
struct MfArgs {
    const float* awm2;     // [Qpad/32][2][16][64][4]: -2 wd o a in MFMA fragment order (see pairmf_prep)
    const float* qA;       // [Qpad]
    const float* b;        // [G][256]
    const float* w;        // [2][256]
    const float* tau;      // [Qpad]   (filter pass)
    float* dense;          // [Qpad][NS_ROWS]   (sample pass)
    float2* region;        // [Qpad][nblocks][rs]  (d', bank row as int bits): slots of (query, GEMM block), written by one wave
    unsigned* rcount;      // [Qpad][nblocks]      used slots (written unconditionally by the owning wave: no memset needed)
    float2* cand;          // [Qpad][cap]  overflow lists (d', bank row as int bits)
    unsigned* cnt;         // [Qpad]       overflow counts
    unsigned* rmax;        // [1] max_j R_j = sum_k (|W0_k| + |W1_k|) b_jk^2 as float bits (R >= 0: uint order == float order)
    int Q, G, cap, rows_per_block, rs, nblocks;
};

template <bool SAMPLE>
__global__ __launch_bounds__(512) void pairmf_kernel(const MfArgs p) {
    __shared__ __attribute__((aligned(16))) float tile[2][TR * LROW];
    __shared__ __attribute__((aligned(16))) float Bn[2][TR];
    __shared__ float wdl[D], awl[D];
    __shared__ float rred[8];
    // candidates a wave finds are parked here (ballot + prefix count, no atomics, no memory round trip beside the MFMAs) and go
    // to the slots of (query, this block) when the block's rows are done: (d', bank row, query)
    __shared__ float wb_v[SAMPLE ? 1 : 8][SAMPLE ? 1 : WCAP];
    __shared__ int wb_g[SAMPLE ? 1 : 8][SAMPLE ? 1 : WCAP];
    __shared__ int wb_q[SAMPLE ? 1 : 8][SAMPLE ? 1 : WCAP];
    __shared__ unsigned qc[8][32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 15, h = lane >> 4;
    const int q0 = blockIdx.y * QG + wid * 32;
    int wcount = 0;                                 // wave-uniform: fill of this wave's LDS buffer
    if (!SAMPLE && lane < 32) qc[wid][lane] = 0u;   // candidates so far of each of the wave's 32 queries (this block)
    auto flush = [&]() {
        if (SAMPLE) return;
        for (int i = lane; i < wcount; i += 64) {
            const int q = wb_q[wid][i];
            const unsigned pos = atomicAdd(&qc[wid][q & 31], 1u);          // wave-private LDS counter
            if (pos < (unsigned)p.rs) {
                p.region[((size_t)q * p.nblocks + blockIdx.x) * p.rs + pos] = make_float2(wb_v[wid][i], __int_as_float(wb_g[wid][i]));
            } else {                                // this (query, block) pair is full (clustered bank): the contended path
                const unsigned slot = atomicAdd(&p.cnt[q], 1u);
                if (slot < (unsigned)p.cap) p.cand[(size_t)q * p.cap + slot] = make_float2(wb_v[wid][i], __int_as_float(wb_g[wid][i]));
            }
        }
        wcount = 0;
    };

    int row_begin, row_end;
    if (SAMPLE) {                                   // block bx samples ONE full 16-row group; groups spread evenly over the bank
        const int nt = p.G / 16, nsb = NS_ROWS / 16;
        row_begin = 16 * (int)(((long)blockIdx.x * nt) / nsb);
        row_end = row_begin + 16;
    } else {
        row_begin = blockIdx.x * p.rows_per_block;
        row_end = min(p.G, row_begin + p.rows_per_block);
        if (row_begin >= p.G) return;
    }
    if (tid < D) {
        const float w0 = p.w[tid], w1 = p.w[D + tid];
        wdl[tid] = w1 - w0;
        awl[tid] = fabsf(w0) + fabsf(w1);
    }

    // ---- bank tile loader: thread -> (row tid >> 4, four 16-byte columns (tid & 15) * 4 + 64 i); 16 threads read 256 B runs
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    f32x4 tv[4];
    auto gload = [&](int r0) {
        const int g = r0 + lr;
        const float* src = p.b + (size_t)g * D + lc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g < row_end) v = *reinterpret_cast<const f32x4*>(src + 64 * i);
            tv[i] = v;
        }
    };
    float rloc = 0.f;
    auto lstore = [&](int buf) {
        float sb = 0.f, sr = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&tile[buf][lr * LROW + lc + 64 * i]) = tv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v2 = tv[i][e] * tv[i][e];
                sb = fmaf(wdl[lc + 64 * i + e], v2, sb);
                sr = fmaf(awl[lc + 64 * i + e], v2, sr);
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {         // the 16 threads of a row are 16 consecutive lanes
            sb += __shfl_xor(sb, o, 64);
            sr += __shfl_xor(sr, o, 64);
        }
        if ((tid & 15) == 0) {
            Bn[buf][lr] = sb;
            rloc = fmaxf(rloc, sr);
        }
    };

    const int ntiles = (row_end - row_begin + TR - 1) / TR;
    gload(row_begin);
    // ---- this wave's 32 queries: B-operand fragments for all 64 k-steps of both column tiles, resident in registers
    // (issued after the first bank tile's loads, so both latencies overlap; each load is a contiguous 1 KiB per wave)
    float bq[2][64];
    float Aq[2], tq[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int q = q0 + 16 * n + j;
        const float* src = p.awm2 + ((size_t)((q0 >> 5) * 2 + n) * 16 * 64 + lane) * 4;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)u * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) bq[n][4 * u + e] = v[e];
        }
        Aq[n] = p.qA[q];
        tq[n] = SAMPLE ? 0.f : p.tau[q];
    }
    __syncthreads();                                // wdl / awl
    lstore(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, r0 = row_begin + t * TR;
        if (t + 1 < ntiles) gload(r0 + TR);
        const bool two = row_end - r0 > 16;         // the last tile of a block may hold one 16-row group only
        f32x4 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* t0 = &tile[buf][j * LROW + 64 * h];
        if (two) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(t0 + 4 * u);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(t0 + 16 * LROW + 4 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], bq[0][4 * u + e], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], bq[1][4 * u + e], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], bq[0][4 * u + e], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], bq[1][4 * u + e], acc[1][1], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(t0 + 4 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], bq[0][4 * u + e], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], bq[1][4 * u + e], acc[0][1], 0, 0, 0);
                }
            }
        }
        // the next tile goes to the other LDS buffer now (its loads were issued before this tile's MFMAs), so that the end of the
        // tile is epilogue + barrier only
        if (t + 1 < ntiles) lstore(buf ^ 1);
        // ---- epilogue: lane holds column (query) j of each column tile and rows 4 h .. 4 h + 3 of each row group
        float vv[16];
        unsigned pm = 0;                            // which of this lane's 16 values pass their query's threshold
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f32x4 bn = *reinterpret_cast<const f32x4*>(&Bn[buf][16 * m + 4 * h]);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = (m * 2 + n) * 4 + r, g = r0 + 16 * m + 4 * h + r;
                    vv[i] = (acc[m][n][r] + Aq[n]) + bn[r];
                    if (SAMPLE) {
                        if (m == 0 || two) p.dense[(size_t)(q0 + 16 * n + j) * NS_ROWS + blockIdx.x * 16 + 4 * h + r] = vv[i];
                    } else if ((m == 0 || two) && g < row_end && vv[i] >= tq[n]) {
                        pm |= 1u << i;
                    }
                }
        }
        if (!SAMPLE && __ballot(pm != 0u) != 0ull) {
            // exclusive prefix count over the wave's lanes, then every lane parks its own candidates
            const int mine = __popc(pm);
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            const int total = __shfl(incl, 63, 64);
            if (wcount + total > WCAP) flush();
            if (total <= WCAP) {
                int pos = wcount + incl - mine;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (pm & (1u << i)) {
                        const int m = i >> 3, n = (i >> 2) & 1, r = i & 3;
                        wb_v[wid][pos] = vv[i]; wb_g[wid][pos] = r0 + 16 * m + 4 * h + r; wb_q[wid][pos] = q0 + 16 * n + j;
                        ++pos;
                    }
                wcount += total;
            } else {                                // degenerate data (most of a tile passes): straight to the lists
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (pm & (1u << i)) {
                        const int m = i >> 3, n = (i >> 2) & 1, r = i & 3, q = q0 + 16 * n + j;
                        const unsigned slot = atomicAdd(&p.cnt[q], 1u);
                        if (slot < (unsigned)p.cap)
                            p.cand[(size_t)q * p.cap + slot] = make_float2(vv[i], __int_as_float(r0 + 16 * m + 4 * h + r));
                    }
            }
        }
        __syncthreads();
    }
    flush();
    if (!SAMPLE && lane < 32) p.rcount[(size_t)(q0 + lane) * p.nblocks + blockIdx.x] = min(qc[wid][lane], (unsigned)p.rs);
    if (!SAMPLE && blockIdx.y == 0) {               // max_j R_j for the error bound (one atomic per block)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rloc = fmaxf(rloc, __shfl_xor(rloc, o, 64));
        if (lane == 0) rred[wid] = rloc;
        __syncthreads();
        if (tid == 0) {
            float r = rred[0];
            for (int i = 1; i < 8; ++i) r = fmaxf(r, rred[i]);
            atomicMax(p.rmax, __float_as_uint(r));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- 3. thresholds
__global__ __launch_bounds__(256) void pairmf_thresh(const float* __restrict__ dense, float* __restrict__ tau, int Q, int kk) {
    __shared__ TopkShared sh;
    __shared__ unsigned surv[4][PAIRMF_KK_MAX];     // kk survivors per wave (kk = pairmf_kk(k) <= PAIRMF_KK_MAX, checked by the launchers)
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (q >= Q) {                                   // padding queries never collect candidates
        if (tid == 0) tau[q] = INFINITY;
        return;
    }
    // two levels, one barrier: each wave takes a quarter of the row (16 keys per lane, in registers) and leaves ITS kk largest keys
    // in LDS (the kk-th largest of the row is among the union); wave 0 then takes the kk-th largest of those 4 kk keys
    constexpr int PER = NS_ROWS / 256;
    unsigned key[PER];
    const float* row = dense + (size_t)q * NS_ROWS + (size_t)wid * (NS_ROWS / 4);
#pragma unroll
    for (int i = 0; i < PER / 4; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + (size_t)(i * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) key[4 * i + e] = tk_key_d(v[e]);
    }
    // 256 histogram words per wave: the hist / key / item / gidx arrays of TopkShared, one each
    unsigned* myhist = wid == 0 ? sh.hist : wid == 1 ? sh.key : wid == 2 ? reinterpret_cast<unsigned*>(sh.item) : reinterpret_cast<unsigned*>(sh.gidx);
    unsigned krem;
    const unsigned T = wave_kth<PER>(key, NS_ROWS / 4, kk, myhist, krem);
    // survivors: every key above T, then krem copies of T (only the VALUES matter for a threshold)
    int base = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const bool up = key[i] > T;
        const unsigned long long mk = __ballot(up);
        if (up) surv[wid][base + __popcll(mk & ((1ull << lane) - 1ull))] = key[i];
        base += __popcll(mk);
    }
    if (lane < (int)krem) surv[wid][base + lane] = T;     // base + krem == kk
    __syncthreads();
    if (wid == 0) {
        constexpr int P2 = (4 * PAIRMF_KK_MAX + 63) / 64;       // keys per lane of the second level: all 4 kk survivors
        unsigned k2[P2];
#pragma unroll
        for (int i = 0; i < P2; ++i) {
            const int e = i * 64 + lane;
            k2[i] = e < 4 * kk ? surv[e / kk][e % kk] : 0u;
        }
        unsigned kr2;
        const unsigned T2 = wave_kth<P2>(k2, 4 * kk, kk, sh.hist, kr2);
        if (lane == 0) tau[q] = key_to_float(T2);
    }
}

// ---------------------------------------------------------------------------------------------------------------- 5. final
struct FinalArgs {
    const float* a; const float* b; const float* w; const float* bias;
    const float2* region; const unsigned* rcount;
    const float2* cand; const unsigned* cnt; const float* tau; const float* qP; const unsigned* rmax;
    int64_t* idx; float* score; int* stats;
    int Q, G, cap, k, kk, force_exact, rs, nblocks;
};

// the direct form, in the operation order of seam_pair_logits_f32 (pair_tile: d = a - b; d2 = d * d; x_c = fma(d2, W_c[k], x_c),
// k ascending from 0; + bias last) -- bit-identical logits
__device__ __forceinline__ void exact_pair(const float* __restrict__ qa, const float* __restrict__ brow, int bstride,
                                           const float* __restrict__ w0, const float* __restrict__ w1, float b0, float b1,
                                           float& x0, float& x1) {
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < D; ++k) {
        const float d = qa[k] - brow[k * bstride];
        const float d2 = d * d;
        acc0 = __builtin_fmaf(d2, w0[k], acc0);
        acc1 = __builtin_fmaf(d2, w1[k], acc1);
    }
    x0 = acc0 + b0;
    x1 = acc1 + b1;
}

__global__ __launch_bounds__(256) void pairmf_final(const FinalArgs p) {
    __shared__ TopkShared sh;
    __shared__ __attribute__((aligned(16))) float qa[D], w0l[D], w1l[D];
    __shared__ __attribute__((aligned(16))) float rows[RB * RLD];
    __shared__ float ex0[256], ex1[256];
    __shared__ float2 lst[LCAP];
    __shared__ unsigned nl_s;
    __shared__ int bad_s;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int k = p.k, kk = p.kk;
    qa[tid] = p.a[(size_t)q * D + tid];
    w0l[tid] = p.w[tid];
    w1l[tid] = p.w[D + tid];
    const float b0 = p.bias[0], b1 = p.bias[1];
    const float tau_q = p.tau[q], qP_q = p.qP[q], rmax = __uint_as_float(p.rmax[0]);      // for the proof, off the critical path
    // ---- gather this query's candidates into LDS: its group's regions (one per GEMM block) + its overflow list
    if (tid == 0) { nl_s = 0u; bad_s = 0; }
    __syncthreads();
    const unsigned over = p.cnt[q];
    if (!p.force_exact) {
        const float2* reg = p.region + (size_t)q * p.nblocks * p.rs;
        const unsigned* rc = p.rcount + (size_t)q * p.nblocks;
        for (int bx = tid; bx < p.nblocks; bx += 256) {
            const int c = (int)rc[bx];
            const float2* e = reg + (size_t)bx * p.rs;
            for (int i = 0; i < c; ++i) {
                const unsigned pos = atomicAdd(&nl_s, 1u);
                if (pos < (unsigned)LCAP) lst[pos] = e[i];
            }
        }
        const int no = (int)min(over, (unsigned)p.cap);
        for (int i = tid; i < no; i += 256) {
            const unsigned pos = atomicAdd(&nl_s, 1u);
            if (pos < (unsigned)LCAP) lst[pos] = p.cand[(size_t)q * p.cap + i];
        }
    }
    __syncthreads();
    const unsigned total = nl_s;
    const int n = (int)min(total, (unsigned)LCAP);
    if (tid == 0) bad_s = (over > (unsigned)p.cap || total > (unsigned)LCAP || n < kk || p.force_exact) ? 1 : 0;
    __syncthreads();
    int64_t* idx_out = p.idx + (size_t)q * k;
    float* score_out = p.score + (size_t)q * k;
    if (!bad_s) {
        auto load = [&](int i, float& x0, float& x1, int& g) { const float2 v = lst[i]; x0 = 0.f; x1 = v.x; g = __float_as_int(v.y); };
        unsigned T, krem;
        block_kth(load, n, kk, sh, T, krem);
        block_collect(load, n, kk, T, krem, sh);      // sh.gidx[0, kk): the kk best candidates by d'
        __syncthreads();
        for (int base = 0; base < kk; base += RB) {
            const int nb = min(RB, kk - base);
            // squared differences of the batch, all threads: rows[r][k] = (a_k - b_rk)^2
            for (int i = tid; i < nb * (D / 4); i += 256) {
                const int r = i >> 6, c4 = i & 63;
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.b + (size_t)sh.gidx[base + r] * D + 4 * c4);
                const f32x4 av = *reinterpret_cast<const f32x4*>(&qa[4 * c4]);
                f32x4 d2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = av[e] - v[e];
                    d2[e] = d * d;
                }
                *reinterpret_cast<f32x4*>(&rows[r * RLD + 4 * c4]) = d2;
            }
            __syncthreads();
            if (tid < nb) {                           // the two fma chains of one pair, k ascending (one thread each: the order is the contract)
                float acc0 = 0.f, acc1 = 0.f;
                const float* dr = rows + tid * RLD;
#pragma unroll 4
                for (int c4 = 0; c4 < D / 4; ++c4) {
                    const f32x4 d2 = *reinterpret_cast<const f32x4*>(dr + 4 * c4);
                    const f32x4 wa = *reinterpret_cast<const f32x4*>(&w0l[4 * c4]);
                    const f32x4 wb = *reinterpret_cast<const f32x4*>(&w1l[4 * c4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_fmaf(d2[e], wa[e], acc0);
                        acc1 = __builtin_fmaf(d2[e], wb[e], acc1);
                    }
                }
                ex0[base + tid] = acc0 + b0;
                ex1[base + tid] = acc1 + b1;
            }
            __syncthreads();
        }
        // order the kk re-scored candidates under the evaluator's order; the first k are the answer
        unsigned mykey = 0;
        int myg = 0, rank = kk;
        if (tid < kk) {
            mykey = tk_key(ex0[tid], ex1[tid]);
            myg = sh.gidx[tid];
        }
        __syncthreads();
        if (tid < kk) sh.key[tid] = mykey;            // sh.key held the d' keys: now the exact ones
        __syncthreads();
        if (tid < kk) {
            rank = block_winner_rank(sh, tid, kk);
            if (rank < k) {
                idx_out[rank] = (int64_t)myg;
                score_out[rank] = tk_score(ex0[tid], ex1[tid]);
            }
            if (rank == k - 1) {
                // Proof that nothing outside the kk candidates belongs to the top k: an outsider j has d'_j <= m (the kk-th best
                // d', itself >= tau_q), and |d'_j - d_j| <= eps_q by the standard fp32 chain bound applied to both evaluations
                // (gamma x the sums of absolute values of their terms <= (sqrt P_q + sqrt R_j)^2 by Cauchy-Schwarz, + the
                // biases); so d_j <= m + eps < d(k-th winner) strictly.  Any NaN makes the comparison false.
                const float m = fmaxf(key_to_float(T), tau_q);
                const float sp = sqrtf(qP_q) + sqrtf(rmax);
                const float eps = 2.f * GAMMA * (sp * sp + fabsf(b0) + fabsf(b1)) + ABS_EPS;
                const float dk = key_to_float(mykey);
                if (!(dk > m + eps)) bad_s = 1;
            }
        }
        __syncthreads();
    }
    if (p.stats && tid == 0) {
        atomicMax(&p.stats[1], (int)min(total, 0x7fffffffu));
        if (over > (unsigned)p.cap || total > (unsigned)LCAP) atomicAdd(&p.stats[2], 1);
        if (bad_s) atomicAdd(&p.stats[0], 1);
    }
    if (bad_s) {
        // this query alone, the direct form over the whole bank (rare: ties / degenerate data at the k-th place)
        const float* bb = p.b;
        block_topk([&](int jj, float& x0, float& x1, int& g) {
            exact_pair(qa, bb + (size_t)jj * D, 1, w0l, w1l, b0, b1, x0, x1);
            g = jj;
        }, p.G, k, idx_out, score_out, sh);
    }
}

int g_num_cu = 0;

}  // namespace

extern "C" {

int seam_pair_topk_mfma_min_gallery(void) { return 2 * NS_ROWS; }
int seam_pair_topk_mfma_max_k(void) { return PAIRMF_MAX_K; }

static int pairmf_kk(int k) { return ((k + 12 + 7) / 8) * 8; }
static_assert(((PAIRMF_MAX_K + 12 + 7) / 8) * 8 <= PAIRMF_KK_MAX, "pairmf_thresh sizes its survivor lists by PAIRMF_KK_MAX");
static int pairmf_cap(int G, int kk) { (void)G; (void)kk; return 1024; }      // overflow list of a query
static int pairmf_rows_per_block(int G) {
    int rpb = (G + g_num_cu - 1) / g_num_cu;
    return (rpb + 15) / 16 * 16;
}
// slots of one (query, GEMM block) pair: 4 x the expectation rows x kk / NS_ROWS, at least 4, a multiple of 2
static int pairmf_region(int rpb, int kk) {
    long e = (4L * rpb * kk + NS_ROWS - 1) / NS_ROWS;
    e = (e + 1) / 2 * 2;
    return (int)(e < 4 ? 4 : (e > 64 ? 64 : e));
}
static void pairmf_cu_count() {
    if (g_num_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        g_num_cu = n;
    }
}

int64_t seam_pair_topk_mfma_workspace_floats(int Q, int G, int k) {
    pairmf_cu_count();
    const int64_t qpad = (Q + QG - 1) / QG * QG;
    const int kk = pairmf_kk(k), rpb = pairmf_rows_per_block(G);
    const int64_t nblocks = (G + rpb - 1) / rpb;
    return qpad * D + 4 * qpad + 16 + qpad * NS_ROWS + 2 * qpad * pairmf_cap(G, kk) + qpad * nblocks * (2 * pairmf_region(rpb, kk) + 1) + 64;
}

// a [Q,256], b [G,256], w [2,256], bias [2] -> idx [Q,k] int64, score [Q,k]: the same values, bit for bit, as
// seam_pair_logits_f32 + seam_rank_topk_f32.  G >= seam_pair_topk_mfma_min_gallery(), k <= seam_pair_topk_mfma_max_k().
// flags bit 0: take the direct-form path for every query (test hook).  stats (device int[4], may be null):
// [0] queries that took the direct-form path, [1] largest candidate count of a query, [2] queries whose list overflowed.
int seam_pair_topk_mfma_f32(const float* a, const float* b, const float* w, const float* bias, int64_t* idx, float* score,
                            int Q, int G, int Dd, int k, float* ws, int flags, int* stats, void* stream) {
    if (Q <= 0 || k <= 0) return 0;
    if (Dd != D || G < 2 * NS_ROWS || k > 64 || k > G || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)ws & 15))
        return (int)hipErrorInvalidValue;
    pairmf_cu_count();
    hipStream_t st = (hipStream_t)stream;
    const int qpad = (Q + QG - 1) / QG * QG;
    const int kk = pairmf_kk(k), cap = pairmf_cap(G, kk);
    float* q = ws;
    float* awm2 = q; q += (size_t)qpad * D;
    float* qA = q; q += qpad;
    float* qP = q; q += qpad;
    float* tau = q; q += qpad;
    unsigned* cnt = reinterpret_cast<unsigned*>(q); q += qpad;
    unsigned* rmax = reinterpret_cast<unsigned*>(q); q += 16;
    float* dense = q; q += (size_t)qpad * NS_ROWS;
    float2* cand = reinterpret_cast<float2*>(q); q += 2 * (size_t)qpad * cap;
    const int rpb = pairmf_rows_per_block(G), nblocks = (G + rpb - 1) / rpb, rs = pairmf_region(rpb, kk);
    float2* region = reinterpret_cast<float2*>(q); q += 2 * (size_t)qpad * nblocks * rs;
    unsigned* rcount = reinterpret_cast<unsigned*>(q);

    hipLaunchKernelGGL(pairmf_prep, dim3(qpad), dim3(256), 0, st, a, w, bias, awm2, qA, qP, cnt, rmax, stats, Q);
    MfArgs m;
    m.awm2 = awm2; m.qA = qA; m.b = b; m.w = w; m.tau = tau; m.dense = dense; m.cand = cand; m.cnt = cnt; m.rmax = rmax;
    m.Q = Q; m.G = G; m.cap = cap; m.region = region; m.rcount = rcount; m.rs = rs; m.nblocks = nblocks;
    const int groups = qpad / QG;
    m.rows_per_block = rpb;
    hipLaunchKernelGGL(pairmf_kernel<true>, dim3(NS_ROWS / 16, groups), dim3(512), 0, st, m);
    hipLaunchKernelGGL(pairmf_thresh, dim3(qpad), dim3(256), 0, st, dense, tau, Q, kk);
    hipLaunchKernelGGL(pairmf_kernel<false>, dim3(nblocks, groups), dim3(512), 0, st, m);
    FinalArgs f;
    f.a = a; f.b = b; f.w = w; f.bias = bias; f.cand = cand; f.cnt = cnt; f.tau = tau; f.qP = qP; f.rmax = rmax;
    f.region = region; f.rcount = rcount; f.rs = rs; f.nblocks = nblocks;
    f.idx = idx; f.score = score; f.stats = stats; f.Q = Q; f.G = G; f.cap = cap; f.k = k; f.kk = kk; f.force_exact = flags & 1;
    hipLaunchKernelGGL(pairmf_final, dim3(Q), dim3(256), 0, st, f);
    return (int)hipGetLastError();
}

}  // extern "C"
