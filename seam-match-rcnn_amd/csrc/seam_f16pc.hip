// seam_f16pc.hip -- 3x3 / stride-1 convolution with fp16 operands and fp32 accumulation (v_mfma_f32_32x32x16_f16) for the
// config-5 path: the direct convolution as a PRODUCER / CONSUMER block, persistent over its XCD's tiles (round 5).
//
// Why a second fp16 kernel.  conv_igemm<_Float16,256,128> runs these layers at 0.35 of the 2.5 PFLOP/s roof (matrix pipe busy
// 0.44, profiles/r05_f16_pmc_traffic.json): every one of the nine taps gathers its own [256 pixels x 64 channels] A tile from
// global memory -- the same input pixel nine times, at a 1-KiB stride -- through the same in-order vmcnt queue as the weights, and
// the eight waves of a block meet at a barrier per tap and chunk.  Here:
//   * a block owns 256 output pixels x 128 output channels: a 8 x 32 patch of a large map, or G whole small maps (the ROI-sized
//     layers: 14x14, 10x10, 8x8, 6x6 outputs);
//   * waves 4..7 (producers) stage the INPUT PATCH -- (rows + 2) x (columns + 2) pixels x 64 channels, one full 128-byte line per
//     pixel, eight adjacent lanes per line -- into LDS once per 64-channel chunk (double buffered, one chunk ahead in LDS, one more
//     in registers), across tile boundaries; they never touch the vector ALU in steady state;
//   * waves 0..3 (consumers, one per SIMD) only multiply: wave w owns output channels 32 w .. 32 w + 31 of the block for ALL 256
//     pixels (8 accumulator tiles = 128 registers).  Per MFMA one `ds_read_b128` takes the A fragment of one 32-pixel group at
//     one tap straight out of the patch (address = the lane's pixel + an immediate tap offset: the nine taps re-read LDS, not
//     memory); per eight MFMAs one 1-KiB global load takes the wave's own B fragment (weights packed in fragment order) through
//     a register ring.  One barrier per chunk (288 MFMAs per wave = 9216 cycles).
//   * epilogue: accumulators (pixels in lanes, four consecutive channels per register quad: the MFMA's operand roles are swapped)
//     -> fp32 exchange rows [pixel][128 channels] in LDS, 64 pixels per pass -> all 512 threads: scale / shift (+ residual),
//     ReLU, fp16, 16-byte NHWC stores.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kOob = 0x80000000u;
constexpr int PXB = 144;                    // LDS bytes per patch pixel: 128 (64 fp16 channels) + 16 -- an odd number of 16-byte slots,
                                            // so the 16 lanes of a ds_read_b128 phase (consecutive pixels) hit 16 different bank groups
constexpr int NPIXMAX = 352;                // patch pixels per buffer (8 x 32 outputs: 10 x 34 = 340)
constexpr int PBUF = NPIXMAX * PXB;         // 50688
constexpr int NP = NPIXMAX * 8 / 256;       // 16-byte pieces per producer thread and chunk: 11
constexpr int EXROW = 512 + 16;             // exchange row: 128 fp32 channels + 16 (33 slots: odd)
constexpr int EXPIX = 64;                   // pixels per epilogue pass
constexpr int EX = 2 * PBUF;                // LDS map: patch[2] | exchange
constexpr int LDS_BYTES = EX + EXPIX * EXROW;
static_assert(LDS_BYTES <= 160 * 1024, "LDS map");
constexpr int RB = 12;                      // B fragments in flight per consumer wave (36 steps per chunk: the ring's phase repeats every chunk)

#define LDSQ __attribute__((address_space(3)))
#define F16_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)

struct F16Args {
    const void* x;         // [N, H, W, C] fp16
    const void* w;         // packed: [K/128][4 n-tiles][C/64 chunks][9 taps][4 k-steps][64 lanes][8 fp16]
    const float* scale;    // [K] or null
    const float* shift;    // [K] or null
    const void* res;       // [N, Ho, Wo, K] fp16 or null
    void* y;               // [N, Ho, Wo, K] fp16
    int N, H, W, C, K, pad, relu;
    int Ho, Wo;
    int mode;              // 0: 8 x 32 output patches of one image; 1: G whole images per block (Ho * Wo * G <= 256)
    int G;
    int PWi, PHi;          // input patch columns / rows per image slot
    int npix;              // patch pixels per block (<= NPIXMAX)
    int bx, by;            // mode 0: patches per image along x / y
    int tiles_m, tiles_n, nchunks, total_tiles;
    unsigned m_tiles_n, m_bx, m_per_img, m_PWi, m_HoWo, m_Wo, m_slotpix;
    int per_img;           // mode 0: bx * by
};

__device__ __forceinline__ int fdivu(int a, int d, unsigned m) { return d == 1 ? a : (int)__umulhi((unsigned)a, m); }

// wave-uniform geometry of one tile
struct F16Geo { int tn, img0, n_here, y0, x0; };
__device__ __forceinline__ F16Geo f16_geo(const F16Args& p, const int tile) {
    F16Geo g;
    const int tm = fdivu(tile, p.tiles_n, p.m_tiles_n);
    g.tn = tile - tm * p.tiles_n;
    if (p.mode == 0) {
        const int img = fdivu(tm, p.per_img, p.m_per_img);
        const int rb = tm - img * p.per_img;
        const int byi = fdivu(rb, p.bx, p.m_bx);
        g.img0 = img; g.n_here = 1;
        g.y0 = byi * 8; g.x0 = (rb - byi * p.bx) * 32;
    } else {
        g.img0 = tm * p.G; g.n_here = min(p.G, p.N - g.img0);
        g.y0 = 0; g.x0 = 0;
    }
    return g;
}
// output slot o (0..255) of a block -> image slot, output row / column inside the patch
__device__ __forceinline__ void f16_slot(const F16Args& p, const int o, int& g, int& oy, int& ox) {
    if (p.mode == 0) { g = 0; oy = o >> 5; ox = o & 31; return; }
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
    float* const ex = reinterpret_cast<float*>(smem + EX);

    // second half of the epilogue for one pass (64 output slots x 128 channels): every thread finishes two 8-channel pieces
    auto finish = [&](const F16Geo& q, const int pass) {
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((char*)p.y + (size_t)q.img0 * out_img), 0, (int)(out_img * q.n_here), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)(p.res ? p.res : p.y) + (size_t)q.img0 * out_img), 0, (int)(out_img * q.n_here), 0x00020000);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int piece = tid + 512 * it;              // 0..1023: slot-in-pass = piece >> 4, 8-channel group = piece & 15
            const int sp = piece >> 4, cg = piece & 15;
            const int o = pass * EXPIX + sp;
            int g, oy, ox;
            f16_slot(p, o, g, oy, ox);
            const int gy = q.y0 + oy, gx = q.x0 + ox;
            const bool ok = g < q.n_here && gy < p.Ho && gx < p.Wo && (p.mode == 0 || o < p.G * p.Ho * p.Wo);
            const int ncol = q.tn * 128 + cg * 8;
            const unsigned off = ok ? (unsigned)(__mul24(__mul24(__mul24(g, p.Ho) + gy, p.Wo) + gx, p.K) + ncol) * 2u : kOob;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ex) + sp * EXROW + cg * 32);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ex) + sp * EXROW + cg * 32 + 16);
            f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
            if (p.scale) { s0 = *reinterpret_cast<const f32x4*>(p.scale + ncol); s1 = *reinterpret_cast<const f32x4*>(p.scale + ncol + 4); }
            if (p.shift) { h0 = *reinterpret_cast<const f32x4*>(p.shift + ncol); h1 = *reinterpret_cast<const f32x4*>(p.shift + ncol + 4); }
            f32x4 a0 = v0 * s0 + h0, a1 = v1 * s1 + h1;
            if (p.res) {
                const f16x8 rv = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) { a0[e] += (float)rv[e]; a1[e] += (float)rv[e + 4]; }
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { a0[e] = fmaxf(a0[e], 0.f); a1[e] = fmaxf(a1[e], 0.f); }
            }
            f16x8 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hv[e] = (_Float16)a0[e]; hv[e + 4] = (_Float16)a1[e]; }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), y_rsrc, off, 0, 0);
        }
    };

    if (!consumer) {
        // =================================================== producer ===================================================
        const int ptid = tid - 256;
        unsigned goff[NP];                      // global byte offset of piece (ptid & 7) of patch pixel (ptid >> 3) + 32 r; kOob outside
        LDSQ char* lp[NP];                      // its LDS address inside patch buffer 0
        auto setup = [&](const F16Geo& q) {     // (vector ALU, beside fp16 MFMAs: the partner's VALU instructions do issue there)
            const int slotpix = p.PHi * p.PWi;
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int pix = (ptid >> 3) + 32 * r;
                const int g = p.mode ? fdivu(pix, slotpix, p.m_slotpix) : 0;
                const int rm = pix - g * slotpix;
                const int iy = fdivu(rm, p.PWi, p.m_PWi);
                const int ix = rm - iy * p.PWi;
                const int gy = q.y0 + iy - p.pad, gx = q.x0 + ix - p.pad;
                const bool inb = pix < p.npix && g < q.n_here && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                goff[r] = inb ? (unsigned)(__mul24(__mul24(__mul24(g, p.H) + gy, p.W) + gx, p.C) * 2 + (ptid & 7) * 16) : kOob;
                lp[r] = (LDSQ char*)smem + (pix < NPIXMAX ? pix * PXB + (ptid & 7) * 16 : 0);
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
                if ((ptid >> 3) + 32 * r < NPIXMAX) *reinterpret_cast<f32x4 LDSQ*>(lp[r] + buf * PBUF) = src[r];
        };
        // Chunk stream of the block: global chunk c = (tile index in the block's walk) * n + chunk-in-tile.  LDS buffer c & 1 holds
        // chunk c while the consumers multiply it; during that time chunk c + 1 goes from registers to the other buffer and chunk
        // c + 2 is requested.  Chunks past a tile's end are the next tile's first ones (their address set is computed when the
        // request stage gets there).
        int tile = tile0, ck = 0;               // the tile / chunk the REQUEST stage is at
        int tiles_left = ntiles;
        F16Geo q = f16_geo(p, tile);
        setup(q);
        __amdgpu_buffer_rsrc_t rs = x_desc(q);
        auto request = [&](f32x4 (&dst)[NP]) {
            if (tiles_left > 0) load_chunk(dst, rs, ck);
            if (++ck == n) {                    // the next request belongs to the next tile
                ck = 0;
                tile += S;
                if (--tiles_left > 0) {
                    q = f16_geo(p, tile);
                    setup(q);
                    rs = x_desc(q);
                }
            }
        };
        request(rq[0]);                         // chunk 0
        request(rq[1]);                         // chunk 1
        store_chunk(rq[0], 0);
        request(rq[0]);                         // chunk 2
        F16_BAR();                              // P: chunk 0 visible
        const int total_chunks = ntiles * n;
        int c = 0;
        for (int k = 0; k < ntiles; ++k) {
            for (int t = 0; t < n; t += 2) {    // two chunks per trip: the register sets' parity is a compile-time constant
                // chunk c (even position in the tile): chunk c + 1 registers (set 1) -> buffer 1; request chunk c + 3 into set 1
                if (c + 1 < total_chunks) store_chunk(rq[1], 1);
                request(rq[1]);
                F16_BAR();
                ++c;
                if (c + 1 < total_chunks) store_chunk(rq[0], 0);
                request(rq[0]);
                F16_BAR();
                ++c;
            }
            // the tile's epilogue: four passes, two barriers each (exchange written / exchange free)
            const F16Geo qe = f16_geo(p, tile0 + k * S);
#pragma unroll 1
            for (int pass = 0; pass < 256 / EXPIX; ++pass) {
                F16_BAR();
                finish(qe, pass);
                F16_BAR();
            }
        }
    } else {
        // =================================================== consumer ===================================================
        const int wn = wave;                    // this wave's 32-channel n-tile of the block's 128
        f32x16 acc[8];
        // per-lane LDS address of output slot 32 m + (lane & 31) at tap (0, 0), k-half (lane >> 5): constant for the whole launch
        LDSQ char* ab[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            int g, oy, ox;
            f16_slot(p, 32 * m + (lane & 31), g, oy, ox);
            if (p.mode && 32 * m + (lane & 31) >= p.G * p.Ho * p.Wo) { g = 0; oy = 0; ox = 0; }      // idle slots read pixel 0 (never stored)
            ab[m] = (LDSQ char*)smem + (__mul24(g, p.PHi * PWI) + oy * PWI + ox) * PXB + (lane >> 5) * 16;
        }
        const int wchunk_bytes = 9 * 4 * 1024;                      // one chunk of one n-tile: 9 taps x 4 k-steps x 1 KiB
        const int wtile_bytes = n * wchunk_bytes;
        const int blane = lane * 16;
        f32x4 af[8];                            // A fragments: the one of pixel group m at the current step, refilled right behind its MFMA
        f32x4 bf[RB];                           // B fragments: step s in slot s % RB
        int tile = tile0;
        F16_BAR();                              // P
        for (int k = 0; k < ntiles; ++k) {
            const F16Geo q = f16_geo(p, tile);
            const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)p.w + (size_t)(q.tn * 4 + wn) * wtile_bytes), 0, wtile_bytes, 0x00020000);
            auto load_b = [&](const int slot, const int step) {     // step = global step of the tile: chunk * 36 + tap * 4 + ks
                bf[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, blane, step * 1024, 0));
            };
#pragma unroll
            for (int s = 0; s < RB; ++s) { SB(); load_b(s, s); }
            SB();
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            for (int t = 0; t < n; ++t) {
                // this chunk's patch buffer: the lanes' pixel addresses move by one buffer (8 adds per 288 MFMAs)
                LDSQ char* ac[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) ac[m] = ab[m] + (t & 1) * PBUF;
                auto read_a = [&](const int m, const int st) -> f32x4 {      // step st = tap * 4 + ks: immediate offset
                    return *reinterpret_cast<const f32x4 LDSQ*>(ac[m] + (((st >> 2) / 3) * PWI + (st >> 2) % 3) * PXB + (st & 3) * 32);
                };
#pragma unroll
                for (int m = 0; m < 8; ++m) af[m] = read_a(m, 0);
#pragma unroll
                for (int st = 0; st < 36; ++st) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        SB();
                        // roles swapped: rows = output channels (the B fragment), columns = pixels (the A fragment)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[st % RB]), __builtin_bit_cast(f16x8, af[m]), acc[m], 0, 0, 0);
                        SB();
                        if (st + 1 < 36) af[m] = read_a(m, st + 1);      // the same pixel group's fragment of the next step
                    }
                    SB();
                    load_b(st % RB, t * 36 + st + RB);                  // past the tile's end: zero fill, never used
                }
                SB();
                F16_BAR();                      // chunk t + 1 is in the other buffer; this one may be overwritten
            }
            // ---- epilogue: four passes of 64 output slots (pixel groups 2 pass, 2 pass + 1) through the fp32 exchange rows ----
#pragma unroll
            for (int pass = 0; pass < 256 / EXPIX; ++pass) {
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const int m = 2 * pass + mm;
                    char* row = reinterpret_cast<char*>(ex) + (32 * mm + (lane & 31)) * EXROW + wn * 128 + (lane >> 5) * 16;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)      // registers 4 qd .. 4 qd + 3 = channels 8 qd + 4 (lane >> 5) + 0..3 of this wave's 32
                        *reinterpret_cast<f32x4*>(row + qd * 32) = f32x4{acc[m][4 * qd], acc[m][4 * qd + 1], acc[m][4 * qd + 2], acc[m][4 * qd + 3]};
                }
                F16_BAR();
                finish(q, pass);
                F16_BAR();
            }
            tile += S;
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
    if (N <= 0 || C < 64 || (C % 64) || K < 128 || (K % 128) || pad < 0 || pad > 1) return 1;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.pad = pad;
    a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2;
    if (a.Ho <= 0 || a.Wo <= 0) return 1;
    a.tiles_n = K / 128;
    a.nchunks = C / 64;
    if (a.nchunks & 1) return 1;
    if (a.Wo <= 16 && a.Ho <= 16) {
        a.mode = 1;
        a.PWi = a.Wo + 2; a.PHi = a.Ho + 2;
        if (a.PWi != 16 && a.PWi != 14 && a.PWi != 12 && a.PWi != 10 && a.PWi != 8) return 1;
        int g = 256 / (a.Ho * a.Wo);
        while (g > 1 && g * a.PHi * a.PWi > NPIXMAX) --g;
        if (g < 1 || a.PHi * a.PWi > NPIXMAX) return 1;
        a.G = g;
        a.npix = g * a.PHi * a.PWi;
        a.tiles_m = (N + g - 1) / g;
        a.bx = a.by = a.per_img = 1;
        if ((size_t)g * H * W * C * 2 >= kOob || (size_t)g * a.Ho * a.Wo * K * 2 >= kOob) return 1;
    } else {
        if (a.Wo < 24) return 1;                // a map too narrow for 32-column patches and too large for the whole-map form
        a.mode = 0; a.G = 1;
        a.PWi = 34; a.PHi = 10; a.npix = 340;
        a.bx = (a.Wo + 31) / 32; a.by = (a.Ho + 7) / 8;
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
    hipLaunchKernelGGL(conv3x3_f16pc<PWI>, dim3(grid), dim3(512), LDS_BYTES, st, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

/* 1 when seam_conv3x3_f16pc takes this layer shape (3x3, stride 1, pad 0 | 1, C a multiple of 128, K a multiple of 128, maps of
 * >= 24 output columns or whole maps of <= 16 x 16 outputs with 16 / 14 / 12 / 10 / 8 input columns), else 0 */
int seam_conv3x3_f16pc_supported(int N, int H, int W, int C, int K, int pad) {
    F16Args a;
    return f16pc_plan(a, N, H, W, C, K, pad) == 0 ? 1 : 0;
}

long long seam_f16pc_weight_halves(int K, int Cstore) { return (long long)K * Cstore * 9; }

int seam_pack_conv_weight_f16pc(const float* w, void* w_packed, int K, int Cin, int Cstore, void* stream) {
    if (K % 128 || Cstore % 64 || Cin > Cstore) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)(K / 32) * (Cstore / 64) * 36 * 64;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(f16pc_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)w_packed, K, Cin, Cstore);
    return (int)hipGetLastError();
}

int seam_conv3x3_f16pc(const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                       int N, int H, int W, int C, int K, int pad, int relu, void* stream) {
    F16Args a;
    if (f16pc_plan(a, N, H, W, C, K, pad)) return (int)hipErrorInvalidValue;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y; a.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    switch (a.PWi) {
        case 34: return f16pc_launch<34>(a, st);
        case 16: return f16pc_launch<16>(a, st);
        case 14: return f16pc_launch<14>(a, st);
        case 12: return f16pc_launch<12>(a, st);
        case 10: return f16pc_launch<10>(a, st);
        case 8: return f16pc_launch<8>(a, st);
    }
    return (int)hipErrorInvalidValue;
}

}  // extern "C"
