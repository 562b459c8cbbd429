// seam_pwpc.hip -- pointwise (1x1, stride 1) convolution with a LONG reduction (C >= 256, a multiple of 128) on the gfx950 fp32
// matrix cores, as a PRODUCER / CONSUMER block persistent over its XCD's tiles (round 5): the bottleneck
// reductions of ResNet layer2-4 (512 / 1024 / 2048 -> 128 / 256 / 512), the layer4 expansions (512 -> 2048).  Exact fp32
// (v_mfma_f32_32x32x2_f32), same contract as seam_conv2d_f32 on those shapes.
//
// Why a third GEMM kernel.  conv1x1_sw (seam_pw.hip) keeps its weight slab in LDS: a longer reduction's slab does not fit.  The
// implicit GEMM (seam_conv.hip) runs these layers at 0.78-0.85 of the fp32-MFMA roof (matrix pipe busy 0.87): its four waves
// stage both operands through LDS themselves and meet at a barrier per 32-channel chunk.  With a long reduction the epilogue is
// noise (1 % of a tile), so what is left to remove is everything in the K loop that is not an MFMA -- the lesson of
// conv3x3_wino24pc / conv3x3_f16pc, without a transform: here the producers never touch the vector ALU at all (the rule of
// beside an fp32-MFMA-saturated wave its SIMD partner's LDS / memory instructions are free, its VALU
// instructions are not issued).
//   * a block owns 128 pixels x 128 output channels per tile; waves 4..7 (producers) copy the tile's activation rows, 64 channels
//     (one 256-byte run per pixel, 16 adjacent lanes per run) per chunk, global -> registers -> LDS (THREE LDS buffers, two more
//     chunks in registers), across tile boundaries; their per-lane offsets are launch invariants, a tile changes one descriptor;
//   * waves 0..3 (consumers, one per SIMD) only multiply: wave w owns channels 32 w .. 32 w + 31 of the tile for all 128 pixels
//     (4 accumulator tiles).  Per 8 k: four `ds_read_b128` (the A fragments of the four pixel groups: lane = pixel, 16 bytes = 4
//     consecutive k, the upper lanes 4 k further) and one 1-KiB global load (the wave's B fragment: weights packed in fragment
//     order, an 8-deep register ring) feed 16 MFMAs; addresses are per-lane constants + immediates: no VALU in the K loop.  One
//     `s_barrier` per chunk (128 MFMAs per wave = 8192 cycles), in the MIDDLE of the chunk: the consumers run from chunk to chunk
//     -- and from a tile's epilogue into the next tile -- without a stop;
//   * the MFMA's operand roles are swapped (rows = channels, columns = pixels), so a lane ends with four consecutive channels
//     of one pixel per register quad; the epilogue transposes each 32 x 32 accumulator tile through a wave-private LDS buffer
//     into rows (8 lanes per pixel) and applies scale / shift, residual, ReLU there: 16-byte stores of full 128-byte lines, no
//     barrier.  Residual pieces are requested before the tile's last chunk.
// Measured (profiles/r05_pwpc_ab.txt, r05_pwpc_trace.txt): 1.01-1.05x the implicit GEMM on the >= 50 x 50 maps it is given, level
// on the 25 x 25 maps (3-6 tiles per CU: both kernels quantise alike), results bit-identical to it on every tested shape (the same
// k order); a chunk takes 8.5-8.9 k cycles for 8192 of MFMA issue, a tile's epilogue 3.4 k.  What the stamps showed on the way:
// 16-byte stores straight from the accumulator layout (32 quarter lines each) took 6.2 k cycles per tile to ISSUE; epilogue loads
// inside `if (scale)` blocks and a dead ring load at a tile's end each left hipcc's vmcnt bookkeeping waiting for the epilogue's
// own earlier stores; a 16-byte store with an SGPR offset followed at once by the next piece's v_pk_fma stored the next piece's
// fourth channel in its last lanes (the store reads its data registers over several cycles; with an immediate offset hipcc adds
// the wait state itself).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include "seam_opts.h"
#if defined(SEAM_PWPC_TRACE)
#include "dev/seam_trace_host.h"      // -DSEAM_DEV_BUILD experiment builds only (tools/experiments/pwpc_abl.sh)
#endif
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128;                     // pixels per tile
constexpr int ROWB = 256 + 16;              // LDS bytes per pixel and chunk: 64 fp32 channels + 16 (17 slots of 16 bytes: odd, so the 16
                                            // lanes of a ds_read_b128 cycle -- 16 different pixels mod 16 -- meet 16 different bank groups)
constexpr int ABUF = BM * ROWB;             // 34816
constexpr int NP = BM * 16 / 256;           // 16-byte pieces per producer thread and chunk: 8
constexpr int TROW = 128 + 16;               // wave-private transpose rows: 32 pixels x (32 channels + 16 bytes)
constexpr int TBUF = 32 * TROW;             // 4608 bytes per consumer wave
constexpr int NBUF = 3;                     // chunk buffers: the barrier sits in the MIDDLE of a chunk (see the kernel)
constexpr int TR0 = NBUF * ABUF;            // LDS map: chunk buffers | per consumer wave two transpose buffers
constexpr int LDS_BYTES = TR0 + 4 * 2 * TBUF;
static_assert(LDS_BYTES <= 160 * 1024, "LDS map");
constexpr int BS = 3;                       // the chunk's barrier follows the MFMAs of this 8-k step
constexpr int RB = 8;                       // B fragments in flight per consumer wave: one chunk (8 steps of 8 k) ahead
constexpr unsigned kOob = 0x80000000u;

#ifndef SEAM_PWPC_ABL
#define SEAM_PWPC_ABL 0     // experiments: 1 producers idle (no loads / LDS stores), 2 no in-loop A fragment reads, 4 no in-loop B fragment loads
#endif
#define LDSQ __attribute__((address_space(3)))
#define PWPC_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)
#ifdef SEAM_PWPC_TRACE
#define PW_TR(tag) do { if (tr_on) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if (lane == 0 && tr_k < 1024) p.trace[wave * 1024 + tr_k] = tm_ | ((unsigned long long)(tag) << 56); ++tr_k; } } while (0)
#else
#define PW_TR(tag) do { } while (0)
#endif

struct PwpcArgs {
    const float* x;        // [M, C]
    const float* w;        // packed: [K/128][4 n-tiles][C/8 steps][64 lanes][4]
    const float* scale;    // [K] or null
    const float* shift;    // [K] or null
    const float* res;      // [M, K] or null
    float* y;              // [M, K]
    int M, C, K, relu;
    int tiles_n, nchunks, total_tiles;
    unsigned m_tiles_n;
    unsigned long long* trace;   // SEAM_PWPC_TRACE builds only
};

__device__ __forceinline__ int fdivu(int a, int d, unsigned m) { return d == 1 ? a : (int)__umulhi((unsigned)a, m); }

__global__ __launch_bounds__(512, 2) void conv1x1_pc(const PwpcArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int n = p.nchunks;

    // ---- the block's tiles: XCD x (= blockIdx & 7) owns a contiguous range of the launch's tiles (tile = m-tile * tiles_n + n-tile:
    // the n-tiles of an activation row block are neighbours); its blocks walk it interleaved ----
    const int T = p.total_tiles, G = gridDim.x;
    const int xcd = blockIdx.x & 7, sl0 = blockIdx.x >> 3;
    const int q8 = T >> 3, rem8 = T & 7;
    const int cnt = q8 + (xcd < rem8 ? 1 : 0);
    const int start = xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8;
    const int S = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
    const int ntiles = sl0 < cnt ? (cnt - sl0 + S - 1) / S : 0;
    if (ntiles == 0) return;
    const int tile0 = start + sl0;
    const size_t row_bytes = (size_t)p.C * 4, out_row = (size_t)p.K * 4;
#ifdef SEAM_PWPC_TRACE
    const bool tr_on = p.trace && blockIdx.x == SEAM_PWPC_TRACE && (wave & 3) == 0;
    int tr_k = 0;
#endif

    if (!consumer) {
        // =================================================== producer ===================================================
        const int ptid = tid - 256;
        unsigned goff[NP];                      // byte offset of piece (ptid & 15) of tile row (ptid >> 4) + 16 r: launch invariants
        LDSQ char* lp[NP];                      // LDS address of the piece in buffer 0 (buffer 1: + ABUF as an immediate) ...
        LDSQ char* lp2[NP];                     // ... and in buffer 2 (past the 16-bit immediate: its own base, no address arithmetic in the loop)
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const int row = (ptid >> 4) + 16 * r;
            goff[r] = (unsigned)(row * p.C * 4 + (ptid & 15) * 16);
            lp[r] = (LDSQ char*)smem + row * ROWB + (ptid & 15) * 16;
            lp2[r] = lp[r] + 2 * ABUF;
            asm volatile("" : "+v"(lp2[r]));    // (kept as a register: hipcc otherwise re-derives it with a v_add per store)
        }
        auto x_desc = [&](const int tile) {     // the tile's rows: rows past M read as zeros (their outputs are never stored)
            const int tm = fdivu(tile, p.tiles_n, p.m_tiles_n);
            const int rows = min(BM, p.M - tm * BM);
            return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + (size_t)tm * BM * row_bytes), 0, (int)(rows * row_bytes), 0x00020000);
        };
        f32x4 rq[2][NP];
        auto load_chunk = [&](f32x4 (&dst)[NP], const __amdgpu_buffer_rsrc_t& rs, const int chunk) {
#pragma unroll
            for (int r = 0; r < NP; ++r) dst[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[r], chunk * 256, 0));
        };
        // chunk c lives in LDS buffer c % 3 and travels through register set c & 1.  The barrier of chunk c (B_c) is passed by the
        // consumers in the MIDDLE of chunk c: before it the producers have stored chunk c + 1 (the consumers flow from chunk to chunk
        // without a stop, reading the next chunk's first fragments ahead), after it chunk c - 1 is history and its buffer takes
        // chunk c + 2.  (Two buffers with the barrier at the chunk's end cost 390 of 8580 cycles per chunk: the barrier's skew plus
        // the LDS latency of the first fragments.)
        auto store_chunk = [&](const f32x4 (&src)[NP], auto buf_c) {
            constexpr int B = decltype(buf_c)::value;
#pragma unroll
            for (int r = 0; r < NP; ++r) *reinterpret_cast<f32x4 LDSQ*>(B == 2 ? lp2[r] : lp[r] + B * ABUF) = src[r];
        };
        int tile = tile0, ck = 0, tiles_left = ntiles;      // the tile / chunk the REQUEST stage is at
        __amdgpu_buffer_rsrc_t rs = x_desc(tile);
        auto request = [&](f32x4 (&dst)[NP]) {
            if (tiles_left > 0) load_chunk(dst, rs, ck);
            if (++ck == n) {
                ck = 0;
                tile += S;
                if (--tiles_left > 0) rs = x_desc(tile);
            }
        };
        const int total_chunks = ntiles * n;
        request(rq[0]);                                     // chunk 0
        request(rq[1]);                                     // chunk 1
        store_chunk(rq[0], std::integral_constant<int, 0>{});
        request(rq[0]);                                     // chunk 2
        store_chunk(rq[1], std::integral_constant<int, 1>{});
        request(rq[1]);                                     // chunk 3
        PWPC_BAR();                                         // P: chunks 0 and 1 visible
        auto step = [&](const int c, auto par_c, auto buf_c) {      // B_c, then chunk c + 2 -> buffer (c + 2) % 3, chunk c + 4 requested
            constexpr int PAR = decltype(par_c)::value;
            PW_TR(11);
            PWPC_BAR();
            PW_TR(12);
#if !(SEAM_PWPC_ABL & 1)
            if (c + 2 < total_chunks) store_chunk(rq[PAR], buf_c);
            request(rq[PAR]);
#endif
        };
        for (int c = 0; c < total_chunks; c += 6) {        // (an even number of chunks per tile: the six phases of (c & 1, c % 3))
            step(c, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            step(c + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            if (c + 2 >= total_chunks) break;
            step(c + 2, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            step(c + 3, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
            if (c + 4 >= total_chunks) break;
            step(c + 4, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            step(c + 5, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        }
    } else {
        // =================================================== consumer ===================================================
        const int wn = wave;
        f32x16 acc[4];
        f32x4 af[4], bf[RB];
        // per-lane LDS address of pixel 32 m + (lane & 31), k-half (lane >> 5), in buffer 0
        const LDSQ char* const ab = (const LDSQ char*)smem + (lane & 31) * ROWB + (lane >> 5) * 16;
        const int nsteps = p.C >> 3;
        const int wtile_bytes = nsteps * 1024;              // one n-tile of one 128-channel block
        const int blane = lane * 16;
        __amdgpu_buffer_rsrc_t w_rsrc;
        int tm = 0, tn = 0;
        auto load_b = [&](const int slot, const int step) {
            bf[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, blane, step * 1024, 0));
        };
        auto ring_preload = [&](const int tl) {
            tm = fdivu(tl, p.tiles_n, p.m_tiles_n);
            tn = __builtin_amdgcn_readfirstlane(tl - tm * p.tiles_n);
            w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w + (size_t)(tn * 4 + wn) * wtile_bytes), 0, wtile_bytes, 0x00020000);
#pragma unroll
            for (int s = 0; s < RB; ++s) { SB(); load_b(s, s); }
            SB();
        };
        // output / residual offsets of this lane inside a tile's row block: row (32 m + lane & 31), channels 32 wn + 4 (lane >> 5) + 8 qd
        // The epilogue's row layout (after the wave-private transpose): lane -> pixel (lane >> 3) + 8 i of a 32-pixel group, 16-byte piece
        // (lane & 7) of the wave's 32 channels: eight lanes write one pixel's 128 bytes -- a full line per pixel.
        const unsigned ooff = (unsigned)((lane >> 3) * p.K * 4 + (32 * wn + 4 * (lane & 7)) * 4);
        LDSQ char* const tw = (LDSQ char*)smem + TR0 + wn * (2 * TBUF) + (lane & 31) * TROW + (lane >> 5) * 16;   // accumulator layout: piece 2 qd + h
        const LDSQ char* const tr = (const LDSQ char*)smem + TR0 + wn * (2 * TBUF) + (lane >> 3) * TROW + (lane & 7) * 16;
        int tile = tile0;
        ring_preload(tile);
        PWPC_BAR();                             // P: chunks 0 and 1 are in LDS
        int cbuf = 0;                           // LDS buffer of the current chunk (global chunk index % 3)
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const f32x4 LDSQ*>(ab + m * (32 * ROWB));
        for (int k = 0; k < ntiles; ++k) {
            const int tm_k = __builtin_amdgcn_readfirstlane(tm), tn_k = __builtin_amdgcn_readfirstlane(tn);   // (wave-uniform by construction)
            const int rows = min(BM, p.M - tm_k * BM);
            const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((char*)p.y + (size_t)tm_k * BM * out_row), 0, (int)(rows * out_row), 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)(p.res ? p.res : p.y) + (size_t)tm_k * BM * out_row), 0, (int)(rows * out_row), 0x00020000);
            PW_TR(1);
            f32x4 rv[4][4], sc, sh;
            // the tile's epilogue vectors (this lane's four channels in the row layout), requested at its start and without a branch (a
            // missing vector reads the weights instead and is replaced by 1 / 0): loads inside `if (scale)` blocks left hipcc's vmcnt
            // bookkeeping with waits behind every store
            {
                const int ch = tn_k * 128 + 32 * wn + 4 * (lane & 7);
                const f32x4 a = *reinterpret_cast<const f32x4*>((p.scale ? p.scale : p.w) + ch);
                const f32x4 b = *reinterpret_cast<const f32x4*>((p.shift ? p.shift : p.w) + ch);
                sc = p.scale ? a : f32x4{1.f, 1.f, 1.f, 1.f};
                sh = p.shift ? b : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const unsigned ocol = ooff + (unsigned)(tn_k * 512);
            const int K32 = 32 * p.K * 4, K8 = 8 * p.K * 4;     // byte steps of a 32-pixel group / of 8 pixels
            for (int t = 0; t < n; ++t) {
                if (t == n - 1 && p.res) {      // the residual pieces (row layout): in flight under the tile's last chunk
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            // (row step in the vector offset, like the stores; either form is range-checked on gfx950 -- vector + scalar +
                            //  immediate against the descriptor's size, profiles/r06_soffset_probe.txt: rows past `rows` read zero)
                            rv[m][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, ocol + m * K32 + i * K8, 0, 0));
                }
                const int nbuf = cbuf == NBUF - 1 ? 0 : cbuf + 1;
                const LDSQ char* const ac = ab + cbuf * ABUF;
                const LDSQ char* const an = ab + nbuf * ABUF;       // the next chunk (the next tile's first one behind a tile's last)
                cbuf = nbuf;
                auto read_a = [&](const int m, const int st) -> f32x4 {      // st = 8: the next chunk's first fragments
                    return *reinterpret_cast<const f32x4 LDSQ*>((st == 8 ? an : ac + st * 32) + m * (32 * ROWB));
                };
                // (one copy of the chunk with two wave-uniform branches per step: three straight-line copies -- first / middle / last
                // chunk of a tile -- measured 3-5 % SLOWER and spilled)
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    // element j of the two fragments = k 8 st + 4 (lane >> 5) + j; consecutive MFMAs go to different accumulators
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            SB();
                            // roles swapped: rows = output channels (the B fragment), columns = pixels (the A fragment)
                            if (st == 0 && j == 0 && t == 0) {
                                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[st % RB][j], af[m][j], z, 0, 0, 0);
                            } else {
                                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[st % RB][j], af[m][j], acc[m], 0, 0, 0);
                            }
                            SB();
#if !(SEAM_PWPC_ABL & 2)
                            if (j == 3) af[m] = read_a(m, st + 1);          // (the next step's fragment: needed four MFMAs = 256 cycles on)
#endif
                        }
                    }
                    SB();
                    // (no request past the tile's end: a dead load into a ring register would make every epilogue instruction that
                    // reuses the register wait -- in vmcnt order -- for the epilogue's own earlier stores)
#if !(SEAM_PWPC_ABL & 4)
                    if (t * 8 + st + RB < nsteps) load_b(st % RB, t * 8 + st + RB);
#endif
                    if (st == BS) {             // B_c: the next chunk is in LDS, the previous one's buffer is free (see the producers)
                        SB();
                        PW_TR(2);
                        asm volatile("s_barrier" ::: "memory");
                        PW_TR(3);
                    }
                }
                SB();
            }
            // ---- epilogue, in registers ----
            tile += S;
            if (k + 1 < ntiles) ring_preload(tile);         // the next tile's first B fragments: in flight under the arithmetic
            PW_TR(4);
            auto finish_tile = [&](auto res_c, auto relu_c) {    // flags as compile-time constants: four straight-line copies
                constexpr bool RES = decltype(res_c)::value, RELU = decltype(relu_c)::value;
                if (RES) {                      // (one wait for everything that came from memory, before the first store)
#pragma unroll
                    for (int m = 0; m < 4; ++m) asm volatile("" :: "v"(rv[m][0]), "v"(rv[m][1]), "v"(rv[m][2]), "v"(rv[m][3]));
                }
                // accumulator layout (lane = pixel, register quad = 4 consecutive channels) -> this wave's private [32 pixels][32 channels]
                // rows (two buffers: group m + 1 is written before group m is read back) -> row layout: 8 lanes per pixel.  The
                // scattered form -- 16-byte stores straight from the accumulator layout, 32 quarter lines per instruction -- spent
                // 6.2 k cycles per tile issuing its 16 stores (stamps).
                auto put = [&](const int m) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        *reinterpret_cast<f32x4 LDSQ*>(tw + (m & 1) * TBUF + qd * 32) =
                            f32x4{acc[m][4 * qd], acc[m][4 * qd + 1], acc[m][4 * qd + 2], acc[m][4 * qd + 3]};
                };
                put(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (m + 1 < 4) put(m + 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4 v = *reinterpret_cast<const f32x4 LDSQ*>(tr + (m & 1) * TBUF + i * (8 * TROW));
                        v = v * sc + sh;                    // (no scale: sc = 1 -- the same value as v + sh)
                        if (RES) v += rv[m][i];
                        if (RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        // (vector offset + immediate only: with a SCALAR offset hipcc puts the next piece's arithmetic right behind a
                        // 16-byte store without the wait state its data registers need -- measured as wrong fourth channels)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc, ocol + m * K32 + i * K8, 0, 0);
                    }
                }
            };
            if (p.res) {
                if (p.relu) finish_tile(std::true_type{}, std::true_type{}); else finish_tile(std::true_type{}, std::false_type{});
            } else {
                if (p.relu) finish_tile(std::false_type{}, std::true_type{}); else finish_tile(std::false_type{}, std::false_type{});
            }
        }
    }
}

// [K, C] row-major fp32 -> fragments [K/128][4][C/8][64][4]:
//   element (tn, w, step, lane, e) = W[n = 128 tn + 32 w + (lane & 31)][c = 8 step + 4 (lane >> 5) + e]
__global__ void pwpc_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int K, int C) {
    const size_t total = (size_t)K * C / 4;
    const int nsteps = C / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        size_t rest = i >> 6;
        const int step = (int)(rest % nsteps); rest /= nsteps;
        const int nt = (int)rest;                // tn * 4 + w
        const int nn = nt * 32 + (lane & 31);
        const int c = 8 * step + 4 * (lane >> 5);
        *reinterpret_cast<f32x4*>(out + i * 4) = *reinterpret_cast<const f32x4*>(w + (size_t)nn * C + c);
    }
}

inline unsigned magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

int pwpc_plan(PwpcArgs& a, long long M, int C, int K) {
    if (M <= 0 || C < 256 || (C % 128) || K < 128 || (K % 128)) return 1;       // (an even number of 64-channel chunks)
    if ((unsigned long long)BM * C * 4 >= kOob || (unsigned long long)BM * K * 4 >= kOob) return 1;
    const long long tiles_m = (M + BM - 1) / BM;
    const long long total = tiles_m * (K / 128);
    if (total >= (1LL << 24) || M >= (1LL << 31)) return 1;
    a.M = (int)M; a.C = C; a.K = K;
    a.tiles_n = K / 128; a.nchunks = C / 64; a.total_tiles = (int)total;
    a.m_tiles_n = magic(a.tiles_n);
    return 0;
}

}  // namespace

extern "C" {

/* 1 when seam_conv1x1_pc_f32 takes this layer (1x1 / stride 1; C >= 256 and a multiple of 128; K a multiple of 128), else 0 */
int seam_conv1x1_pc_supported(long long M, int C, int K) {
    PwpcArgs a;
    return pwpc_plan(a, M, C, K) == 0 ? 1 : 0;
}

long long seam_conv1x1_pc_weight_floats(int K, int C) { return (long long)K * C; }

int seam_pack_conv1x1_pc_f32(const float* w, float* w_packed, int K, int C, void* stream) {
    if (K % 128 || C % 128) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)K * C / 4;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pwpc_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, w_packed, K, C);
    return (int)hipGetLastError();
}

int seam_conv1x1_pc_f32(const float* x, const float* w_packed, const float* scale, const float* shift, const float* residual, float* y,
                        long long M, int C, int K, int relu, void* stream) {
    PwpcArgs a;
    if (pwpc_plan(a, M, C, K)) return (int)hipErrorInvalidValue;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y; a.relu = relu;
    static std::atomic<unsigned> attr_done{0};
    static std::atomic<int> cus[32];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv1x1_pc, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        cus[dev & 31].store(ncu, std::memory_order_relaxed);
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int ncu = cus[dev & 31].load(std::memory_order_relaxed);
    const unsigned grid = (unsigned)(a.total_tiles > ncu ? ncu : a.total_tiles);
#ifdef SEAM_PWPC_TRACE
    static seam_dev::TraceBuf tb;
    a.trace = seam_dev::trace_begin(tb, 8 * 1024);
    hipLaunchKernelGGL(conv1x1_pc, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    seam_dev::trace_end(tb, 2, 1024, false, 0);
    return (int)hipGetLastError();
#else
    a.trace = nullptr;
    hipLaunchKernelGGL(conv1x1_pc, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
#endif
}

}  // extern "C"
