// seam_f16pc.hip -- 3x3 / stride-1 convolution with fp16 operands and fp32 accumulation (v_mfma_f32_32x32x16_f16) for the
// config-5 path: the direct convolution as a PRODUCER / CONSUMER block, persistent over its XCD's tiles (round 5).
//
// Why a second fp16 kernel.  conv_igemm<_Float16,256,128> runs these layers at 0.35 of the 2.5 PFLOP/s roof (matrix pipe busy
// 0.44, profiles/r05_f16_pmc_traffic.json): every one of the nine taps gathers its own [256 pixels x 64 channels] A tile from
// global memory -- the same input pixel nine times, at a 1-KiB stride -- through the same in-order vmcnt queue as the weights, and
// the eight waves of a block meet at a barrier per tap and chunk.  Here:
//   * a block owns up to 256 output pixels x 128 output channels: a 8 x 32 patch of a large map, or G whole small maps (the
//     ROI-sized layers: 14x14, 12x12, 10x10, 8x8, 6x6 outputs) in as many 32-pixel groups as they need (5..7: f16_nm);
//   * waves 4..7 (producers) stage the INPUT PATCH -- (rows + 2) x (columns + 2) pixels x 64 channels, one full 128-byte line per
//     pixel, eight adjacent lanes per line -- into LDS once per 64-channel chunk (double buffered, one chunk ahead in LDS, one more
//     in registers), across tile boundaries; they never touch the vector ALU in steady state;
//   * waves 0..3 (consumers, one per SIMD) only multiply: wave w owns output channels 32 w .. 32 w + 31 of the block for ALL its
//     pixels (up to 8 accumulator tiles = 128 registers).  Per MFMA one `ds_read_b128` takes the A fragment of one 32-pixel group at
//     one tap straight out of the patch (address = the lane's pixel + an immediate tap offset: the nine taps re-read LDS, not
//     memory); per eight MFMAs one 1-KiB global load takes the wave's own B fragment (weights packed in fragment order) through
//     a register ring.  One barrier per chunk (288 MFMAs per wave = 9216 cycles).  The patch rows of the small maps are padded
//     so that the 16 lanes of an LDS cycle meet 16 different bank groups (f16_rowp).
//   * epilogue: the consumers finish their own accumulators in registers (pixels in lanes, four consecutive channels per register
//     quad -- the MFMA's operand roles are swapped; the tile's scale / shift vectors come from a 1-KiB LDS row the producers
//     filled during the last chunk): fp32 scale / shift, fp16 rounding, ReLU, then 8-byte swizzled writes of the whole
//     256 x 128 fp16 tile into LDS -- over the patch buffer the tile's LAST chunk just left, which nobody needs before the next
//     tile's chunk 1 -- ONE barrier, and on to the next tile's first chunk (its first B fragments were requested before the
//     finishing arithmetic).  The producers drain the tile to memory (16-byte NHWC stores) beside that chunk's MFMAs.
//     (The first form went through fp32 exchange rows in four passes of two barriers, all 512 threads finishing: 7.9 k of a
//     tile's 50 k cycles, profiles/r05_f16pc_trace.txt.)  A residual operand is not taken: no 3x3 layer of the path has one.
// Measured (profiles/r05_f16pc_ab.txt, r05_f16pc_pmc.txt, r05_f16pc_ablation.txt): 1.29-1.49x the implicit GEMM on every config-5
// 3x3 shape whose tiles are >= 3/4 full; 192 x 336 x 256 -> 256: 1.2 PFLOP/s, matrix pipe busy 0.845 -- at 1.50 GHz: the 1300 W
// package limit, not the issue rate, bounds it (bare MFMAs out of registers sustain 1.76 PFLOP/s at 1.72 GHz on dense operands).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "seam_opts.h"
#if defined(SEAM_F16PC_TRACE)
#include "dev/seam_trace_host.h"      // -DSEAM_DEV_BUILD experiment builds only (tools/experiments/f16pc_abl.sh)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kOob = 0x80000000u;
constexpr int PXB = 144;                    // LDS bytes per patch pixel: 128 (64 fp16 channels) + 16 -- nine 16-byte slots: consecutive pixels
                                            // fall into consecutive-times-nine bank groups (mod 16: all different)
constexpr int NPIXMAX = 352;                // patch pixels per buffer (8 x 32 outputs: 10 x 34 = 340)
constexpr int PBUF = 56 * 1024;             // bytes per patch buffer (its last 128 are the dump row of the idle producer lanes)
// LDS pitch of a patch ROW in 16-byte slots.  The 16 lanes a ds_read_b128 serves per LDS cycle read 16 different OUTPUT slots o;
// they are conflict free when their pixels' slot numbers differ mod 16.  Large maps: the lanes' pixels are neighbours of one
// row.  Whole small maps: output slot o = oy * Wo + ox sits at patch pixel oy * PWi + ox -- two columns skipped per row -- so the
// row pitch is padded to 9 Wo (mod 16), and the image pitch to 9 Ho Wo: the slot number of output o is 9 o (mod 16) again.
// (Unpadded, the lane groups of the 8 x 8 ... 16 x 16 maps met 1.9 ... 3.0 pixels per bank group: the LDS, serving four consumer
// waves one fragment per MFMA, was the bound of those layers -- tools/experiments/f16pc_banks.py.)
constexpr int f16_rowp(int pwi) { return pwi == 34 ? 34 * 9 : pwi == 18 ? 18 * 9 : 9 * (pwi - 2) + 32; }
constexpr int NP = NPIXMAX * 8 / 256;       // 16-byte pieces per producer thread and chunk: 11
constexpr int EX = PBUF;                    // LDS map: patch[2]; the finished fp16 tile (256 pixels x 256 bytes) overlays patch 1 and beyond;
constexpr int SS = EX + 65536;              // then the tile's scale[128] | shift[128] fp32 row
constexpr int DC = SS + 1024;               // then the producers' drain count
constexpr int LDS_BYTES = DC + 16;
static_assert(2 * PBUF <= SS, "the finished tile must cover patch buffer 1");
static_assert(LDS_BYTES <= 160 * 1024, "LDS map");
// 32-pixel groups of a block's tile as a function of the patch width: 8 (256 output slots) for the large maps; for the whole-map
// form as many as the maps that fit need -- 16 x 16 inputs -> 14 x 14 outputs: 196 slots in 7 groups (8 would idle 23 % of every
// MFMA); 14 -> 12 x 12: 144 in 5; 12 -> 10 x 10: two maps, 200 in 7; 10 -> 8 x 8: three maps, 192 in 6; 8 -> 6 x 6: five maps, 180 in 6
constexpr int f16_nm(int pwi) { return pwi == 34 || pwi == 18 ? 8 : pwi == 16 ? 7 : pwi == 14 ? 5 : pwi == 12 ? 7 : 6; }
// B fragments in flight per consumer wave (a divisor of the 36 steps of a chunk: the ring's phase repeats every chunk); the short
// tiles look further ahead in steps -- the same distance in cycles
constexpr int f16_rb(int nm) { return nm <= 6 ? 18 : 12; }

#ifndef SEAM_F16PC_ABL
#define SEAM_F16PC_ABL 0     // experiments: 1 no in-loop A fragment reads, 2 no in-loop B fragment loads, 4 no patch staging
#endif
#define LDSQ __attribute__((address_space(3)))
#define F16_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)
#ifdef SEAM_F16PC_TRACE
#define F16_TR(tag) do { if (tr_on) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if (lane == 0 && tr_k < 2048) p.trace[wave * 2048 + tr_k] = tm_ | ((unsigned long long)(tag) << 56); ++tr_k; } } while (0)
#else
#define F16_TR(tag) do { } while (0)
#endif

struct F16Args {
    const void* x;         // [N, H, W, C] fp16
    const void* w;         // packed: [K/128][4 n-tiles][C/64 chunks][9 taps][4 k-steps][64 lanes][8 fp16]
    const float* scale;    // [K] or null
    const float* shift;    // [K] or null
    void* y;               // [N, Ho, Wo, K] fp16
    int N, H, W, C, K, pad, relu;
    int Ho, Wo;
    int mode;              // 0: 8 x 32 output patches of one image; 2: 16 x 16 output patches; 1: G whole images per block (Ho * Wo * G <= 256)
    int G;
    int PWi, PHi;          // input patch columns / rows per image slot
    int npix;              // patch pixels per block (<= NPIXMAX)
    int imgp;              // LDS pitch of an image slot's patch in 16-byte slots
    int bx, by;            // mode 0: patches per image along x / y
    int tiles_m, tiles_n, nchunks, total_tiles;
    unsigned m_tiles_n, m_bx, m_per_img, m_PWi, m_HoWo, m_Wo, m_slotpix;
    int per_img;           // mode 0: bx * by
    unsigned long long* trace;   // SEAM_F16PC_TRACE builds only
};

__device__ __forceinline__ int fdivu(int a, int d, unsigned m) { return d == 1 ? a : (int)__umulhi((unsigned)a, m); }

// wave-uniform geometry of one tile
struct F16Geo { int tn, img0, n_here, y0, x0; };
// MODE 0: 8 x 32 output patches of one image; 2: 16 x 16 output patches (round 6: maps that 32-column patches tile badly -- 192 x 336
// leaves 4.5 % of every 8 x 32 launch on columns that do not exist, 96 x 168 12.5 %); 1: G whole small images per block
template <int MODE>
__device__ __forceinline__ F16Geo f16_geo(const F16Args& p, const int tile) {
    F16Geo g;
    const int tm = fdivu(tile, p.tiles_n, p.m_tiles_n);
    g.tn = tile - tm * p.tiles_n;
    if (MODE != 1) {
        const int img = fdivu(tm, p.per_img, p.m_per_img);
        const int rb = tm - img * p.per_img;
        const int byi = fdivu(rb, p.bx, p.m_bx);
        g.img0 = img; g.n_here = 1;
        g.y0 = byi * (MODE == 0 ? 8 : 16); g.x0 = (rb - byi * p.bx) * (MODE == 0 ? 32 : 16);
    } else {
        g.img0 = tm * p.G; g.n_here = min(p.G, p.N - g.img0);
        g.y0 = 0; g.x0 = 0;
    }
    return g;
}
// output slot o (0..255) of a block -> image slot, output row / column inside the patch
template <int MODE>
__device__ __forceinline__ void f16_slot(const F16Args& p, const int o, int& g, int& oy, int& ox) {
    if (MODE == 0) { g = 0; oy = o >> 5; ox = o & 31; return; }
    if (MODE == 2) { g = 0; oy = o >> 4; ox = o & 15; return; }
    const int HoWo = p.Ho * p.Wo;
    g = fdivu(o, HoWo, p.m_HoWo);
    const int rm = o - g * HoWo;
    oy = fdivu(rm, p.Wo, p.m_Wo);
    ox = rm - oy * p.Wo;
}

// PWI: input patch columns as a compile-time constant (34: large maps; 16 / 14 / 12 / 10 / 8: the ROI-sized maps), so that the nine
// tap offsets of an A fragment read are immediates
template <int PWI>
__global__ __launch_bounds__(512, 2) void conv3x3_f16pc(const F16Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    constexpr int NM = f16_nm(PWI), RB = f16_rb(NM), ROWP = f16_rowp(PWI);
    constexpr int MODE = PWI == 34 ? 0 : PWI == 18 ? 2 : 1;      // large maps in 8 x 32 or 16 x 16 patches | G whole small maps per block
    const int n = p.nchunks;

    // ---- the block's tiles: XCD x (= blockIdx & 7) owns a contiguous range of the launch's tiles; its blocks walk it interleaved ----
    const int T = p.total_tiles, G = gridDim.x;
    const int xcd = blockIdx.x & 7, sl0 = blockIdx.x >> 3;
    const int q8 = T >> 3, rem8 = T & 7;
    const int cnt = q8 + (xcd < rem8 ? 1 : 0);
    const int start = xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8;
    const int S = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
    const int ntiles = sl0 < cnt ? (cnt - sl0 + S - 1) / S : 0;
    if (ntiles == 0) return;
    const int tile0 = start + sl0;
    const size_t img_bytes = (size_t)p.H * p.W * p.C * 2;
    const size_t out_img = (size_t)p.Ho * p.Wo * p.K * 2;
#ifdef SEAM_F16PC_TRACE
    const bool tr_on = p.trace && blockIdx.x == SEAM_F16PC_TRACE && (wave & 3) == 0;
    int tr_k = 0;
#endif

    if (!consumer) {
        // =================================================== producer ===================================================
        const int ptid = tid - 256;
        unsigned goff[NP];                      // global byte offset of piece (ptid & 7) of patch pixel (ptid >> 3) + 32 r; kOob outside
        LDSQ char* lp[NP];                      // its LDS address inside patch buffer 0
        {                                       // (launch invariants; lanes past the patch write the buffer's dump row)
            const int slotpix = p.PHi * p.PWi;
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int pix = (ptid >> 3) + 32 * r;
                const int g = MODE == 1 ? fdivu(pix, slotpix, p.m_slotpix) : 0;
                const int rm = pix - g * slotpix;
                const int iy = fdivu(rm, p.PWi, p.m_PWi);
                const int ix = rm - iy * p.PWi;
                const int at = pix < p.npix ? (__mul24(g, p.imgp) + iy * ROWP + ix * 9) * 16 : PBUF - 128;
                lp[r] = (LDSQ char*)smem + at + (ptid & 7) * 16;
            }
        }
        auto setup = [&](const F16Geo& q) {     // (vector ALU, beside fp16 MFMAs: the partner's VALU instructions do issue there)
            const int slotpix = p.PHi * p.PWi;
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int pix = (ptid >> 3) + 32 * r;
                const int g = MODE == 1 ? fdivu(pix, slotpix, p.m_slotpix) : 0;
                const int rm = pix - g * slotpix;
                const int iy = fdivu(rm, p.PWi, p.m_PWi);
                const int ix = rm - iy * p.PWi;
                const int gy = q.y0 + iy - p.pad, gx = q.x0 + ix - p.pad;
                const bool inb = pix < p.npix && g < q.n_here && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                goff[r] = inb ? (unsigned)(__mul24(__mul24(__mul24(g, p.H) + gy, p.W) + gx, p.C) * 2 + (ptid & 7) * 16) : kOob;
            }
        };
        auto x_desc = [&](const F16Geo& q) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + (size_t)q.img0 * img_bytes), 0, (int)(img_bytes * q.n_here), 0x00020000);
        };
        f32x4 rq[2][NP];                        // patch chunks in registers (chunk parity)
        auto load_chunk = [&](f32x4 (&dst)[NP], const __amdgpu_buffer_rsrc_t& rs, const int chunk) {
#pragma unroll
            for (int r = 0; r < NP; ++r)
                dst[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[r], chunk * 128, 0));
        };
        auto store_chunk = [&](const f32x4 (&src)[NP], const int buf) {
#pragma unroll
            for (int r = 0; r < NP; ++r)
                *reinterpret_cast<f32x4 LDSQ*>(lp[r] + buf * PBUF) = src[r];
        };
        // Chunk stream of the block: global chunk c = (tile index in the block's walk) * n + chunk-in-tile.  LDS buffer c & 1 holds
        // chunk c while the consumers multiply it; during that time chunk c + 1 goes from registers to the other buffer and chunk
        // c + 2 is requested.  Chunks past a tile's end are the next tile's first ones (their address set is computed when the
        // request stage gets there).
        int tile = tile0, ck = 0;               // the tile / chunk the REQUEST stage is at
        int tiles_left = ntiles;
        F16Geo q = f16_geo<MODE>(p, tile);
        setup(q);
        __amdgpu_buffer_rsrc_t rs = x_desc(q);
        auto request = [&](f32x4 (&dst)[NP]) {
            if (tiles_left > 0) load_chunk(dst, rs, ck);
            if (++ck == n) {                    // the next request belongs to the next tile
                ck = 0;
                tile += S;
                if (--tiles_left > 0) {
                    q = f16_geo<MODE>(p, tile);
                    setup(q);
                    rs = x_desc(q);
                }
            }
        };
        if (ptid == 0) *reinterpret_cast<LDSQ unsigned*>((LDSQ char*)smem + DC) = 0u;
        request(rq[0]);                         // chunk 0
        request(rq[1]);                         // chunk 1
        store_chunk(rq[0], 0);
        request(rq[0]);                         // chunk 2
        F16_BAR();                              // P: chunk 0 visible
        const int total_chunks = ntiles * n;
        int c = 0;
        for (int k = 0; k < ntiles; ++k) {
            for (int t = 0; t < n; t += 2) {    // two chunks per trip: the register sets' parity is a compile-time constant
                if (t == n - 2 && ptid < 64) {  // the tile's epilogue vectors -> LDS (read by the consumers after the last chunk's barrier)
                    const int i4 = (ptid & 31) * 4;
                    const float* src = ptid < 32 ? p.scale : p.shift;
                    const float dflt = ptid < 32 ? 1.f : 0.f;
                    const int tl = tile0 + k * S;
                    const int tn = tl - fdivu(tl, p.tiles_n, p.m_tiles_n) * p.tiles_n;
                    const f32x4 vv = src ? *reinterpret_cast<const f32x4*>(src + tn * 128 + i4) : f32x4{dflt, dflt, dflt, dflt};
                    *reinterpret_cast<f32x4 LDSQ*>((LDSQ char*)smem + SS + ptid * 16) = vv;
                }
                // chunk c (even position in the tile): chunk c + 1 registers (set 1) -> buffer 1; request chunk c + 3 into set 1
                F16_TR(11);
#if !(SEAM_F16PC_ABL & 4)
                if (c + 1 < total_chunks) store_chunk(rq[1], 1);
                request(rq[1]);
#endif
                F16_TR(12);
                F16_BAR();
                ++c;
#if !(SEAM_F16PC_ABL & 4)
                if (c + 1 < total_chunks) store_chunk(rq[0], 0);
                request(rq[0]);
#endif
                F16_BAR();
                ++c;
            }
            // the tile's epilogue: the consumers' finished fp16 tile -> memory, beside the next tile's first chunk
            F16_BAR();                          // E: the tile is in LDS
            {
                const F16Geo qe = f16_geo<MODE>(p, tile0 + k * S);
                const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)((char*)p.y + (size_t)qe.img0 * out_img), 0, (int)(out_img * qe.n_here), 0x00020000);
                const int piece = ptid & 15;
                const int ncol = qe.tn * 128 + piece * 8;
                int ob = ptid >> 4;
                asm volatile("" : "+v"(ob));    // the rows' output coordinates are launch invariants: computed HERE, not kept (spilled) for the whole kernel
#pragma unroll
                for (int hf = 0; hf < (2 * NM + 3) / 4; ++hf) {
                    u32x4 v[4];
                    SB();
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (hf * 4 + i >= 2 * NM) continue;
                        const int o = (hf * 4 + i) * 16 + ob;
                        v[i] = *reinterpret_cast<const u32x4 LDSQ*>((LDSQ char*)smem + EX + o * 256 + ((piece ^ (o & 15)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int it = hf * 4 + i;
                        if (it >= 2 * NM) continue;
                        const int o = it * 16 + ob;
                        int g, oy, ox;
                        f16_slot<MODE>(p, o, g, oy, ox);
                        const int gy = qe.y0 + oy, gx = qe.x0 + ox;
                        const bool ok = g < qe.n_here && gy < p.Ho && gx < p.Wo && (MODE != 1 || o < p.G * p.Ho * p.Wo);
                        const unsigned off = ok ? (unsigned)(__mul24(__mul24(__mul24(g, p.Ho) + gy, p.Wo) + gx, p.K) + ncol) * 2u : kOob;
                        // rows with bit 4 set keep their two 8-byte halves swapped (the consumers' conflict-free write pattern)
                        const u32x4 w = (it & 1) ? u32x4{v[i][2], v[i][3], v[i][0], v[i][1]} : v[i];
                        __builtin_amdgcn_raw_buffer_store_b128(w, y_rsrc, off, 0, 0);
                    }
                    SB();
                }
            }
            // D: every producer wave is through with the tile's rows before any of them puts the next tile's chunk 1 over them (the
            // four waves drain interleaved rows and store interleaved patch pixels; without D a fast wave's patch stores ran into
            // a slow wave's rows: a handful of wrong pixels per launch).  A count in LDS among the four producer waves -- the
            // consumers are in the middle of their chunk 0 and take no part.  (LDS operations of a wave execute in order: a
            // wave's increment follows its row reads.)
            if (k + 1 < ntiles) {
                LDSQ unsigned* const cnt = reinterpret_cast<LDSQ unsigned*>((LDSQ char*)smem + DC);
                if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned want = 4u * (unsigned)(k + 1);
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(2);
                asm volatile("" ::: "memory");
            }
        }
    } else {
        // =================================================== consumer ===================================================
        const int wn = wave;                    // this wave's 32-channel n-tile of the block's 128
        f32x16 acc[NM];
        // per-lane LDS address of output slot 32 m + (lane & 31) at tap (0, 0), k-half (lane >> 5): constant for the whole launch
        LDSQ char* ab[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            int g, oy, ox;
            f16_slot<MODE>(p, 32 * m + (lane & 31), g, oy, ox);
            if (MODE == 1 && 32 * m + (lane & 31) >= p.G * p.Ho * p.Wo) { g = 0; oy = 0; ox = 0; }      // idle slots read pixel 0 (never stored)
            ab[m] = (LDSQ char*)smem + (__mul24(g, p.imgp) + oy * ROWP + ox * 9) * 16 + (lane >> 5) * 16;
        }
        const int wchunk_bytes = 9 * 4 * 1024;                      // one chunk of one n-tile: 9 taps x 4 k-steps x 1 KiB
        const int wtile_bytes = n * wchunk_bytes;
        const int blane = lane * 16;
        f32x4 af[NM];                            // A fragments: the one of pixel group m at the current step, refilled right behind its MFMA
        f32x4 bf[RB];                           // B fragments: step s in slot s % RB
        int tile = tile0;
        F16_BAR();                              // P
        __amdgpu_buffer_rsrc_t w_rsrc;
        auto load_b = [&](const int slot, const int step) {         // step = global step of the tile: chunk * 36 + tap * 4 + ks
            bf[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, blane, step * 1024, 0));
        };
        auto ring_preload = [&](const int tl) {                     // the first RB fragments of tile tl's own weight slice
            const F16Geo q = f16_geo<MODE>(p, tl);
            w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w + (size_t)(q.tn * 4 + wn) * wtile_bytes), 0, wtile_bytes, 0x00020000);
#pragma unroll
            for (int s = 0; s < RB; ++s) { SB(); load_b(s, s); }
            SB();
        };
        ring_preload(tile);
        for (int k = 0; k < ntiles; ++k) {
            F16_TR(1);
            for (int t = 0; t < n; ++t) {
                // this chunk's patch buffer: the lanes' pixel addresses move by one buffer (8 adds per 288 MFMAs)
                LDSQ char* ac[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) ac[m] = ab[m] + (t & 1) * PBUF;
                auto read_a = [&](const int m, const int st) -> f32x4 {      // step st = tap * 4 + ks: immediate offset
                    return *reinterpret_cast<const f32x4 LDSQ*>(ac[m] + (((st >> 2) / 3) * ROWP + ((st >> 2) % 3) * 9) * 16 + (st & 3) * 32);
                };
#pragma unroll
                for (int m = 0; m < NM; ++m) af[m] = read_a(m, 0);
#pragma unroll
                for (int st = 0; st < 36; ++st) {
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        SB();
                        // roles swapped: rows = output channels (the B fragment), columns = pixels (the A fragment)
                        if (st == 0 && t == 0) {     // the tile's first step multiplies into a constant zero: no accumulator clears
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[st % RB]), __builtin_bit_cast(f16x8, af[m]), z, 0, 0, 0);
                        } else {
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[st % RB]), __builtin_bit_cast(f16x8, af[m]), acc[m], 0, 0, 0);
                        }
                        SB();
#if !(SEAM_F16PC_ABL & 1)
                        if (st + 1 < 36) af[m] = read_a(m, st + 1);      // the same pixel group's fragment of the next step
#endif
                    }
                    SB();
#if !(SEAM_F16PC_ABL & 2)
                    load_b(st % RB, min(t * 36 + st + RB, n * 36 - 1));  // (past the tile's end: its last step again, never used)
#endif
                }
                SB();
                F16_TR(2);
                F16_BAR();                      // chunk t + 1 is in the other buffer; this one may be overwritten
                F16_TR(3);
            }
            // ---- epilogue ----
            tile += S;
            F16_TR(4);
            if (k + 1 < ntiles) ring_preload(tile);     // in flight under the finishing arithmetic
            f32x4 sc[4], sh[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {    // this lane's 16 channels: 8 qd + 4 (lane >> 5) + 0..3 of the wave's 32
                sc[qd] = *reinterpret_cast<const f32x4 LDSQ*>((LDSQ char*)smem + SS + (wn * 32 + 8 * qd + 4 * (lane >> 5)) * 4);
                sh[qd] = *reinterpret_cast<const f32x4 LDSQ*>((LDSQ char*)smem + SS + 512 + (wn * 32 + 8 * qd + 4 * (lane >> 5)) * 4);
            }
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            int le = lane;
            asm volatile("" : "+v"(le));        // the 40 row / piece addresses below are computed here, not kept (spilled) across the chunk loop
            auto finish_tile = [&](auto relu_c) {        // ReLU as a compile-time constant: two copies of the loop, no per-value selects
                constexpr bool RELU = decltype(relu_c)::value;
                const f16x2 lo = f16x2{(_Float16)0.f, (_Float16)0.f};
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int o = 32 * m + (le & 31);
                    // row o: 16 pieces of 16 bytes, piece index XOR (o & 15); rows with bit 4 set swap the 8-byte halves of a piece
                    LDSQ char* row = (LDSQ char*)smem + EX + o * 256 + ((((le >> 5) ^ (o >> 4)) & 1) << 3);
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const f32x2 a0 = f32x2{acc[m][4 * qd], acc[m][4 * qd + 1]} * f32x2{sc[qd][0], sc[qd][1]} + f32x2{sh[qd][0], sh[qd][1]};
                        const f32x2 a1 = f32x2{acc[m][4 * qd + 2], acc[m][4 * qd + 3]} * f32x2{sc[qd][2], sc[qd][3]} + f32x2{sh[qd][2], sh[qd][3]};
                        f16x2 h0 = __builtin_convertvector(a0, f16x2), h1 = __builtin_convertvector(a1, f16x2);
                        if (RELU) {             // after the rounding: the same values as ReLU before it (monotone, 0 exact)
                            h0 = __builtin_elementwise_max(h0, lo);
                            h1 = __builtin_elementwise_max(h1, lo);
                        }
                        u32x2 pk;
                        pk[0] = __builtin_bit_cast(unsigned, h0);
                        pk[1] = __builtin_bit_cast(unsigned, h1);
                        *reinterpret_cast<u32x2 LDSQ*>(row + (((wn * 4 + qd) ^ (o & 15)) << 4)) = pk;
                    }
                }
            };
            if (p.relu) finish_tile(std::true_type{}); else finish_tile(std::false_type{});
            F16_TR(5);
            F16_BAR();                          // E: the producers take it from here
            F16_TR(6);
        }
    }
}

// =====================================================================================================================
// conv3x3_f16pc64 (round 6): C = 64 -> K = 64 (the three 3x3 layers of layer1) -- a tile is ONE 64-channel chunk, so the chunk
// pipeline above has nothing to run ahead on, and the layer is bound by memory (32 KiB in + 32 KiB out per 144 MFMAs of a
// wave), not by the matrix pipe.  Same roles, other proportions:
//   * a tile is 16 x 16 outputs (an 18 x 18 patch: 27 % halo against 33 % for 8 x 32, and the config-5 map -- 192 x 336 -- tiles
//     exactly: 8 x 32 tiles leave 4.5 % of every launch's MFMAs on columns that do not exist);
//   * the WEIGHTS ARE STATIONARY IN REGISTERS: consumer wave w owns output channels 32 (w & 1) .. + 31 for the pixel rows
//     8 (w >> 1) .. + 7 of the tile; its 36 B fragments (9 taps x 4 k-steps x 16 bytes per lane = 144 registers)
//     are loaded once per launch; 64 accumulators; the K loop is 144 MFMAs + 144 ds_read_b128 and nothing else;
//   * the producers keep two tiles' patches in flight in registers and one ahead in LDS (patch[2]), and drain the finished tile
//     (its own 32 KiB LDS region) beside the next tile's MFMAs;
//   * ONE barrier per tile (behind the consumers' epilogue); that the producers have taken tile k - 1 out of the exchange region
//     before the consumers put tile k into it is a count in LDS (it is always there long before: the drain is 8 loads + 8 stores
//     per producer thread at the start of a 4608-cycle interval).
// =====================================================================================================================
#ifndef SEAM_F16PC64_ABL
#define SEAM_F16PC64_ABL 0   // experiments (results are garbage): 1 no in-loop A reads, 2 no epilogue, 4 no patch requests, 8 no drain stores, 16 no MFMAs, 32 no patch LDS stores
#endif
constexpr int PB64 = 46 * 1024;                 // patch buffer: 18 x 18 pixels x 144 bytes + the dump row
constexpr int ROWP64 = 18 * 9;                  // LDS pitch of a patch row in 16-byte slots
constexpr int OUT64 = 2 * PB64;                 // the finished fp16 tile: 256 pixels x 128 bytes
constexpr int SS64 = OUT64 + 32768;             // scale[64] | shift[64] fp32
constexpr int DC64 = SS64 + 512;                // the producers' drain count
constexpr int LDS64 = DC64 + 16;
static_assert(324 * PXB + 128 <= PB64 && LDS64 <= 160 * 1024, "LDS map (C = 64 form)");
__device__ __forceinline__ F16Geo f16_geo64(const F16Args& p, const int tm) {
    F16Geo g;
    const int img = fdivu(tm, p.per_img, p.m_per_img);
    const int rb = tm - img * p.per_img;
    const int byi = fdivu(rb, p.bx, p.m_bx);
    g.tn = 0; g.img0 = img; g.n_here = 1;
    g.y0 = byi * 16; g.x0 = (rb - byi * p.bx) * 16;
    return g;
}

__global__ __launch_bounds__(512, 2) void conv3x3_f16pc64(const F16Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    constexpr int ROWP = ROWP64;

    const int T = p.total_tiles, G = gridDim.x;
    const int xcd = blockIdx.x & 7, sl0 = blockIdx.x >> 3;
    const int q8 = T >> 3, rem8 = T & 7;
    const int cnt = q8 + (xcd < rem8 ? 1 : 0);
    const int start = xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8;
    const int S = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
    const int ntiles = sl0 < cnt ? (cnt - sl0 + S - 1) / S : 0;
    if (ntiles == 0) return;
    const int tile0 = start + sl0;
    const size_t img_bytes = (size_t)p.H * p.W * 64 * 2;
    const size_t out_img = (size_t)p.Ho * p.Wo * 64 * 2;
    LDSQ unsigned* const dcnt = reinterpret_cast<LDSQ unsigned*>((LDSQ char*)smem + DC64);

    if (!consumer) {
        // =================================================== producer ===================================================
        const int ptid = tid - 256;
        unsigned goff[NP];
        LDSQ char* lp[NP];
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const int pix = (ptid >> 3) + 32 * r;
            const int iy = fdivu(pix, 18, p.m_PWi);
            const int ix = pix - iy * 18;
            lp[r] = (LDSQ char*)smem + (pix < 324 ? (iy * ROWP + ix * 9) * 16 : PB64 - 128) + (ptid & 7) * 16;
        }
        auto setup = [&](const F16Geo& q) {
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int pix = (ptid >> 3) + 32 * r;
                const int iy = fdivu(pix, 18, p.m_PWi);
                const int ix = pix - iy * 18;
                const int gy = q.y0 + iy - p.pad, gx = q.x0 + ix - p.pad;
                const bool inb = pix < 324 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                goff[r] = inb ? (unsigned)(__mul24(__mul24(gy, p.W) + gx, 64) * 2 + (ptid & 7) * 16) : kOob;
            }
        };
        f32x4 rq[2][NP];                        // two tiles' patches in registers (tile parity)
        int tile = tile0, tiles_left = ntiles;  // the tile the REQUEST stage is at
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0, 0x00020000);
        // Every interval issues the SAME vector-memory instructions in the same order (8 stores of the drain, 11 loads of the
        // request; without a tile they carry out-of-range offsets and touch nothing), so that hipcc's vmcnt bookkeeping is exact:
        // with conditional requests it merged the paths and waited for the patch requested ONE interval ago -- half the distance.
        auto request = [&](f32x4 (&dst)[NP]) {
            if (tiles_left > 0) {
                const F16Geo q = f16_geo64(p, tile);
                setup(q);
                rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + (size_t)q.img0 * img_bytes), 0, (int)img_bytes, 0x00020000);
                tile += S;
                --tiles_left;
            } else {
#pragma unroll
                for (int r = 0; r < NP; ++r) goff[r] = kOob;
            }
#pragma unroll
            for (int r = 0; r < NP; ++r)
                dst[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (SEAM_F16PC64_ABL & 4) ? kOob : goff[r], 0, 0));
        };
        auto store_patch = [&](const f32x4 (&src)[NP], const int buf) {
            if (SEAM_F16PC64_ABL & 32) return;
#pragma unroll
            for (int r = 0; r < NP; ++r)
                *reinterpret_cast<f32x4 LDSQ*>(lp[r] + buf * PB64) = src[r];
        };
        auto drain = [&](const int k) {         // the finished tile k: LDS -> memory, 128 contiguous bytes per pixel (k < 0: nothing is stored)
            const F16Geo qe = f16_geo64(p, tile0 + max(k, 0) * S);
            const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((char*)p.y + (size_t)qe.img0 * out_img), 0, (int)out_img, 0x00020000);
            const int piece = ptid & 7, ob = ptid >> 3;
            const int sw = (ob >> 1) & 7;
            const bool hs = (ob >> 4) & 1;
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = *reinterpret_cast<const u32x4 LDSQ*>((LDSQ char*)smem + OUT64 + (32 * i + ob) * 128 + ((piece ^ sw) << 4));
#pragma unroll
            for (int i = 0; i < 8; ++i) {     // row o = 32 i + ob of the exchange region = output (2 i + (ob >> 4), ob & 15) of the tile
                const int gy = qe.y0 + 2 * i + (ob >> 4), gx = qe.x0 + (ob & 15);
                const bool ok = k >= 0 && gy < p.Ho && gx < p.Wo && !(SEAM_F16PC64_ABL & 8);
                const unsigned off = ok ? (unsigned)(__mul24(__mul24(gy, p.Wo) + gx, 64) + piece * 8) * 2u : kOob;
                const u32x4 w = hs ? u32x4{v[i][2], v[i][3], v[i][0], v[i][1]} : v[i];
                __builtin_amdgcn_raw_buffer_store_b128(w, y_rsrc, off, 0, 0);
            }
            // the rows are in registers: the consumers may put the next tile over them (interval k + 1 brings the count to 4 (k + 2))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(dcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if (ptid == 0) *dcnt = 0u;
        if (ptid < 32) {                        // the epilogue vectors (one n-tile: the same for every tile of the launch)
            const float* src = ptid < 16 ? p.scale : p.shift;
            const float dflt = ptid < 16 ? 1.f : 0.f;
            const f32x4 vv = src ? *reinterpret_cast<const f32x4*>(src + (ptid & 15) * 4) : f32x4{dflt, dflt, dflt, dflt};
            *reinterpret_cast<f32x4 LDSQ*>((LDSQ char*)smem + SS64 + ptid * 16) = vv;
        }
        request(rq[0]);                         // tile 0
        request(rq[1]);                         // tile 1
        store_patch(rq[0], 0);
        request(rq[0]);                         // tile 2
        F16_BAR();                              // P: tile 0's patch, the epilogue vectors and the count are visible
        // interval k (the consumers multiply tile k out of patch[k & 1]): tile k - 1 out of the exchange region; tile k + 1
        // registers -> patch[(k + 1) & 1]; request tile k + 3 into the freed registers; barrier
        drain(-1);
        store_patch(rq[1], 1);
        request(rq[1]);
        F16_BAR();                              // E(0)
        for (int k = 1; k < ntiles; k += 2) {
            drain(k - 1);
            store_patch(rq[0], 0);              // (past the last tile: zeros into a buffer nobody reads)
            request(rq[0]);
            F16_BAR();                          // E(k)
            if (k + 1 >= ntiles) break;
            drain(k);
            store_patch(rq[1], 1);
            request(rq[1]);
            F16_BAR();                          // E(k + 1)
        }
        drain(ntiles - 1);
    } else {
        // =================================================== consumer ===================================================
        const int wn = wave & 1, ph = wave >> 1;
        f32x4 bf[36];                           // this wave's weights: 32 channels x 64 x 9, for the whole launch
        {
            const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w + (size_t)wn * 36 * 1024), 0, 36 * 1024, 0x00020000);
#pragma unroll
            for (int s = 0; s < 36; ++s)
                bf[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, lane * 16, s * 1024, 0));
        }
        // pixel group m of the wave = tile rows 8 ph + 2 m, + 1 (16 lanes each): the lane's pixel at tap (0, 0), k-half lane >> 5, in
        // patch[0]; groups and taps are immediates
        LDSQ char* const ab0 = (LDSQ char*)smem + ((8 * ph + ((lane >> 4) & 1)) * ROWP + (lane & 15) * 9) * 16 + (lane >> 5) * 16;
        f32x16 acc[4];
        f32x4 af[4];
        F16_BAR();                              // P
        for (int k = 0; k < ntiles; ++k) {
            LDSQ char* const ac = ab0 + (k & 1) * PB64;
            auto read_a = [&](const int m, const int st) -> f32x4 {
                return *reinterpret_cast<const f32x4 LDSQ*>(ac + ((2 * m + (st >> 2) / 3) * ROWP + ((st >> 2) % 3) * 9) * 16 + (st & 3) * 32);
            };
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = read_a(m, 0);
#pragma unroll
            for (int st = 0; st < 36; ++st) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    SB();
                    if (st == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[st]), __builtin_bit_cast(f16x8, af[m]), z, 0, 0, 0);
                    } else if ((SEAM_F16PC64_ABL & 16) && st > 1) {
                    } else {
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[st]), __builtin_bit_cast(f16x8, af[m]), acc[m], 0, 0, 0);
                    }
                    SB();
                    if (st + 1 < 36 && !(SEAM_F16PC64_ABL & 1)) af[m] = read_a(m, st + 1);
                }
            }
            SB();
            // ---- epilogue: fp32 scale / shift, fp16 rounding, ReLU; 8-byte swizzled writes of the lane's pixels into the exchange region ----
            {
                const unsigned want = 4u * (unsigned)(k + 1);       // interval k's drain (of tile k - 1; interval 0: an empty one) is through
                while (__hip_atomic_load(dcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
            }
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const int h = lane >> 5, ol = lane & 31;
            // row o = 32 (4 ph + m) + ol: 8 pieces of 16 bytes at piece ^ ((o >> 1) & 7); rows with bit 4 set swap a piece's halves
            LDSQ char* const row0 = (LDSQ char*)smem + OUT64 + (128 * ph + ol) * 128 + (((h ^ (ol >> 4)) & 1) << 3);
            const int sw = (ol >> 1) & 7;
            auto finish_tile = [&](auto relu_c) {
                constexpr bool RELU = decltype(relu_c)::value;
                const f16x2 lo = f16x2{(_Float16)0.f, (_Float16)0.f};
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x4 sc = *reinterpret_cast<const f32x4 LDSQ*>((LDSQ char*)smem + SS64 + (wn * 32 + 8 * qd + 4 * h) * 4);
                    const f32x4 sh = *reinterpret_cast<const f32x4 LDSQ*>((LDSQ char*)smem + SS64 + 256 + (wn * 32 + 8 * qd + 4 * h) * 4);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const f32x2 a0 = f32x2{acc[m][4 * qd], acc[m][4 * qd + 1]} * f32x2{sc[0], sc[1]} + f32x2{sh[0], sh[1]};
                        const f32x2 a1 = f32x2{acc[m][4 * qd + 2], acc[m][4 * qd + 3]} * f32x2{sc[2], sc[3]} + f32x2{sh[2], sh[3]};
                        f16x2 h0 = __builtin_convertvector(a0, f16x2), h1 = __builtin_convertvector(a1, f16x2);
                        if (RELU) {
                            h0 = __builtin_elementwise_max(h0, lo);
                            h1 = __builtin_elementwise_max(h1, lo);
                        }
                        u32x2 pk;
                        pk[0] = __builtin_bit_cast(unsigned, h0);
                        pk[1] = __builtin_bit_cast(unsigned, h1);
                        *reinterpret_cast<u32x2 LDSQ*>(row0 + m * 32 * 128 + (((wn * 4 + qd) ^ sw) << 4)) = pk;
                    }
                }
            };
            if (SEAM_F16PC64_ABL & 2) { if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123456.75f) *reinterpret_cast<float LDSQ*>(row0) = 1.f; }
            else if (p.relu) finish_tile(std::true_type{}); else finish_tile(std::false_type{});
            F16_BAR();                          // E(k)
        }
    }
}

// OIHW fp32 [K, Cin, 3, 3] -> fp16 fragments [K/128][4][Cs/64][9][4][64][8]:
//   element (tn, w, chunk, tap, ks, lane, e) = W[n = 128 tn + 32 w + (lane & 31)][c = 64 chunk + 16 ks + 8 (lane >> 5) + e][tap / 3][tap % 3]
__global__ void f16pc_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int K, int Cin, int Cs) {
    const int nch = Cs / 64;
    const size_t total = (size_t)(K / 32) * nch * 36 * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        size_t rest = i >> 6;
        const int st = (int)(rest % 36); rest /= 36;
        const int chunk = (int)(rest % nch);
        const int nt32 = (int)(rest / nch);                         // 4 tn + w
        const int tap = st >> 2, ks = st & 3;
        const int nn = nt32 * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = chunk * 64 + ks * 16 + (lane >> 5) * 8 + e;
            const float v = c < Cin ? w[(((size_t)nn * Cin + c) * 3 + tap / 3) * 3 + tap % 3] : 0.f;
            out[i * 8 + e] = (_Float16)v;
        }
    }
}

inline unsigned magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

// fills `a` for a supported shape; 0 = supported
int f16pc_plan(F16Args& a, int N, int H, int W, int C, int K, int pad) {
    const bool c64 = C == 64 && K == 64;          // conv3x3_f16pc64: one chunk, one n-tile, large maps only
    if (N <= 0 || C < 64 || (C % 64) || ((K < 128 || (K % 128)) && !c64) || pad < 0 || pad > 1) return 1;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.pad = pad;
    a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2;
    if (a.Ho <= 0 || a.Wo <= 0) return 1;
    a.tiles_n = c64 ? 1 : K / 128;
    a.nchunks = C / 64;
    if ((a.nchunks & 1) && !c64) return 1;
    if (c64 && a.Wo <= 16 && a.Ho <= 16) return 1;
    if (a.Wo <= 16 && a.Ho <= 16) {
        a.mode = 1;
        a.PWi = a.Wo + 2; a.PHi = a.Ho + 2;
        if (a.PWi != 16 && a.PWi != 14 && a.PWi != 12 && a.PWi != 10 && a.PWi != 8) return 1;
        int g = 32 * f16_nm(a.PWi) / (a.Ho * a.Wo);
        const int rowp = f16_rowp(a.PWi);
        a.imgp = a.PHi * rowp;
        a.imgp += ((9 * a.Ho * a.Wo - a.imgp) % 16 + 16) % 16;
        while (g > 1 && (g * a.PHi * a.PWi > NPIXMAX || g * a.imgp * 16 > PBUF - 128)) --g;
        if (g < 1 || a.PHi * a.PWi > NPIXMAX || a.imgp * 16 > PBUF - 128) return 1;
        a.G = g;
        a.npix = g * a.PHi * a.PWi;
        a.tiles_m = (N + g - 1) / g;
        a.bx = a.by = a.per_img = 1;
        if ((size_t)g * H * W * C * 2 >= kOob || (size_t)g * a.Ho * a.Wo * K * 2 >= kOob) return 1;
    } else {
        if (a.Wo < 24) return 1;                // a map too narrow for 32-column patches and too large for the whole-map form
        a.mode = 0; a.G = 1;
        a.PWi = 34; a.PHi = 10; a.npix = 340; a.imgp = 10 * f16_rowp(34);
        a.bx = (a.Wo + 31) / 32; a.by = (a.Ho + 7) / 8;
        {   // 16 x 16 patches where they leave fewer empty slots (the same sums in the same order: the results do not depend on it)
            const int bx2 = (a.Wo + 15) / 16, by2 = (a.Ho + 15) / 16;
            if (!c64 && bx2 * by2 < a.bx * a.by && seam_opt::get(seam_opt::F16PC_TILE16)) {
                a.mode = 2; a.PWi = 18; a.PHi = 18; a.npix = 324; a.imgp = 18 * f16_rowp(18); a.bx = bx2; a.by = by2;
            }
        }
        if (c64) { a.PWi = 18; a.PHi = 18; a.npix = 324; a.imgp = 18 * ROWP64; a.bx = (a.Wo + 15) / 16; a.by = (a.Ho + 15) / 16; }
        a.per_img = a.bx * a.by;
        a.tiles_m = N * a.per_img;
        if ((size_t)H * W * C * 2 >= kOob || (size_t)a.Ho * a.Wo * K * 2 >= kOob) return 1;
    }
    const long total = (long)a.tiles_m * a.tiles_n;
    if (total >= (1L << 24)) return 1;
    a.total_tiles = (int)total;
    a.m_tiles_n = magic(a.tiles_n); a.m_bx = magic(a.bx); a.m_per_img = magic(a.per_img); a.m_PWi = magic(a.PWi);
    a.m_HoWo = magic(a.Ho * a.Wo); a.m_Wo = magic(a.Wo); a.m_slotpix = magic(a.PHi * a.PWi);
    return 0;
}

template <int PWI>
int f16pc_launch(const F16Args& a, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    static std::atomic<int> cus[32];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv3x3_f16pc<PWI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        cus[dev & 31].store(ncu, std::memory_order_relaxed);
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int ncu = cus[dev & 31].load(std::memory_order_relaxed);
    const unsigned grid = (unsigned)(a.total_tiles > ncu ? ncu : a.total_tiles);
#ifdef SEAM_F16PC_TRACE
    static seam_dev::TraceBuf tb;
    F16Args b = a; b.trace = seam_dev::trace_begin(tb, 8 * 2048);
    hipLaunchKernelGGL(conv3x3_f16pc<PWI>, dim3(grid), dim3(512), LDS_BYTES, st, b);
    seam_dev::trace_end(tb, 2, 2048, false, 0);
    return (int)hipGetLastError();
#else
    hipLaunchKernelGGL(conv3x3_f16pc<PWI>, dim3(grid), dim3(512), LDS_BYTES, st, a);
    return (int)hipGetLastError();
#endif
}

int f16pc64_launch(const F16Args& a, hipStream_t st) {
    static std::atomic<unsigned> attr_done{0};
    static std::atomic<int> cus[32];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv3x3_f16pc64, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64);
        if (e != hipSuccess) return (int)e;
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        cus[dev & 31].store(ncu, std::memory_order_relaxed);
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int ncu = cus[dev & 31].load(std::memory_order_relaxed);
    const unsigned grid = (unsigned)(a.total_tiles > ncu ? ncu : a.total_tiles);
    hipLaunchKernelGGL(conv3x3_f16pc64, dim3(grid), dim3(512), LDS64, st, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

/* 1 when seam_conv3x3_f16pc takes this layer shape (3x3, stride 1, pad 0 | 1, C a multiple of 128 and K a multiple of 128 -- or C = K = 64 on maps of >= 24 output columns --, maps of
 * >= 24 output columns or whole maps of <= 16 x 16 outputs with 16 / 14 / 12 / 10 / 8 input columns), else 0 */
int seam_conv3x3_f16pc_supported(int N, int H, int W, int C, int K, int pad) {
    F16Args a;
    return f16pc_plan(a, N, H, W, C, K, pad) == 0 ? 1 : 0;
}

/* 1 when the kernel is also expected to beat conv_igemm<_Float16,256,128> on this map geometry: its 256-slot tiles must be >= 3/4
 * full (measured, profiles/r05_f16pc_ab.txt: 1.47x at 95 % fill, 1.23-1.38x at 77-88 %, 1.03x at 75 %, 0.87-0.95x at 56-70 %).
 * The batch size takes no part: a frame gives the same bits alone as inside a batch (tests/test_gpu_config5.py). */
int seam_conv3x3_f16pc_pays(int N, int H, int W, int C, int K, int pad) {
    F16Args a;
    if (f16pc_plan(a, N, H, W, C, K, pad)) return 0;
    const double slots = a.mode != 1 ? (double)a.bx * a.by * 256.0 : 32.0 * f16_nm(a.PWi);
    const double used = a.mode != 1 ? (double)a.Ho * a.Wo : (double)a.G * a.Ho * a.Wo;
    return used >= 0.75 * slots ? 1 : 0;
}

long long seam_f16pc_weight_halves(int K, int Cstore) { return (long long)K * Cstore * 9; }

int seam_pack_conv_weight_f16pc(const float* w, void* w_packed, int K, int Cin, int Cstore, void* stream) {
    if ((K % 128 && !(K == 64 && Cstore == 64)) || Cstore % 64 || Cin > Cstore) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)(K / 32) * (Cstore / 64) * 36 * 64;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(f16pc_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)w_packed, K, Cin, Cstore);
    return (int)hipGetLastError();
}

int seam_conv3x3_f16pc(const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                       int N, int H, int W, int C, int K, int pad, int relu, void* stream) {
    F16Args a;
    if (residual || f16pc_plan(a, N, H, W, C, K, pad)) return (int)hipErrorInvalidValue;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y; a.relu = relu;
    a.trace = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (a.K == 64) return f16pc64_launch(a, st);
    switch (a.PWi) {
        case 34: return f16pc_launch<34>(a, st);
        case 18: return f16pc_launch<18>(a, st);
        case 16: return f16pc_launch<16>(a, st);
        case 14: return f16pc_launch<14>(a, st);
        case 12: return f16pc_launch<12>(a, st);
        case 10: return f16pc_launch<10>(a, st);
        case 8: return f16pc_launch<8>(a, st);
    }
    return (int)hipErrorInvalidValue;
}

}  // extern "C"
