// seam_wino.hip -- Winograd F(2x2,3x3) convolution on the gfx950 fp32 matrix cores.
//
// Serves the stride-1 3x3 layers of the path (ResNet-50 bottleneck 3x3s, FPN output convs, RPN head conv, mask head,
// match-trunk valid convs: ~70 % of the extractor's FLOPs) with 2.25x fewer MFMA issues than the implicit GEMM of
// seam_conv.hip:  Y = At [ (G g Gt) (.) (Bt d B) ] A  per 2x2 output tile, summed over input channels, i.e. 16
// independent GEMMs  M_p[tile, n] = sum_c V_p[tile, c] * U_p[n, c]  (p = (xi, nu) in 4x4), everything in fp32
// (v_mfma_f32_32x32x2_f32; the transforms only add/subtract, the weight transform is done once at pack time).
//
// Mapping to CDNA4 (wave64, 4 SIMDs / CU) -- there is NO operand staging through LDS:
//   block = 256 threads = 4 waves; wave xi owns the four positions (xi, nu = 0..3) for TM = 32*MT tiles x 32 output
//   channels: 4 * MT accumulator tiles of 32x32 (64 / 128 VGPRs).
//   A operand: lane (tile = l & 31, khalf = l >> 5) needs V_p[tile][4*khalf .. +3] for its own positions only, so each
//     wave computes row xi of Bt d (two input rows per column) and the four column combinations IN REGISTERS, straight
//     into MFMA fragment layout.  The only LDS traffic is the raw input patch of the block ((2*TY+2) x (2*TX+2) pixels
//     x 8 channels, split by channel half and x-parity so the tile-strided b128 reads are conflict free), double
//     buffered, one barrier per 8-channel chunk.
//   B operand: the transformed weights are packed in fragment order [n_tile][chunk][p][lane][4 floats]; a wave streams
//     its 4 KiB per chunk with coalesced buffer loads directly into registers (L2 resident: every block walks the same
//     chunks), one chunk ahead.
//   Raw patch loads: global -> registers (two sets, issued two chunks ahead) -> LDS; padding and tails are hardware
//     out-of-range zero fills (no branches around loads).
//   Epilogue: the nu half of the output transform in registers, the xi half through a 32 KiB LDS exchange, then
//     scale/shift (+ residual, ReLU) and 16-byte NHWC stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "seam_opts.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned kOob = 0x80000000u;

// Packed-fp32 VALU ops for the input transform.  Written as asm because the DAG combiner scalarises a <4 x float> op
// whose lanes are extracted one by one (each feeds its own MFMA): 32 v_fma/v_sub per transform instead of 16 packed ones.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x4 fma4(f32x2 c, f32x4 b, f32x4 a) {        // a + c * b
    const f32x2 lo = pk_fma(c, __builtin_shufflevector(b, b, 0, 1), __builtin_shufflevector(a, a, 0, 1));
    const f32x2 hi = pk_fma(c, __builtin_shufflevector(b, b, 2, 3), __builtin_shufflevector(a, a, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_add(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_add(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_sub(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_sub(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

#ifndef SEAM_WINO_ABL
#define SEAM_WINO_ABL 0     // kernel experiments (operands keep the REAL data of chunks 0/1 -- zeros would run at a higher clock):
                            // 1 no in-loop patch loads / LDS stores, 2 no in-loop weight loads, 4 no barrier, 8 no in-loop transforms
#endif

struct WinoArgs {
    const float* x;
    const float* u;       // packed transformed weights
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int N, H, W, C, K;
    int Ho, Wo, pad, relu;
    // Up to three regions tile an image: [0] the main region (exact multiple of its block patch), [1] the right strip,
    // [2] the bottom strip -- each with its own patch shape, so edge blocks are not half empty.
    int nreg;
    int rx0[3], ry0[3];   // first tile column / row of the region
    int rxe[3], rye[3];   // one past its last tile column / row
    int TX[3], TY[3];     // tiles per block patch (TX*TY <= 32*MT)
    int bx[3], by[3];     // blocks along x / y
    int per_img;          // blocks per image (sum over regions)
    // Stacked mode (small maps, e.g. the 14x14 ROI tiles of the heads): the tile rows of ALL images form one tall list
    // (row R = n * tiles_y + ty); a block takes TY[0] consecutive rows at full width (TX[0] = tiles_x), crossing image
    // boundaries.  Its raw patch lives in "padded row" space: image n owns rows [n * pitch, (n+1) * pitch), pitch =
    // 2 * tiles_y + 2, padded row r of an image = its input row r - pad.
    int stack;
    int tiles_y;          // tile rows per image
    int PH;               // stacked mode: raw patch rows (max over blocks)
    int G;                // stacked mode: max images a block touches (else 1)
    int tiles_n;          // K / 32
    int nchunks;          // C / 8
    // n-tile split over XCD groups (round 6): nsplit groups of XCDs, tns = tiles_n / nsplit n-tiles per group; the m-blocks are cut
    // into 8 / nsplit partitions of part_q (+1 for the first part_r) each.  nsplit = 1: every XCD walks all n-tiles of its m-blocks.
    int nsplit, tns, part_q, part_r;
};

template <int MT> struct WinoCfg {
    static constexpr int NPIXMAX = MT == 2 ? 384 : 208;           // raw patch pixels per buffer
    static constexpr int NI = (2 * NPIXMAX + 255) / 256;          // 16-byte raw loads per thread per chunk
};

template <int MT>
__global__ __launch_bounds__(256, 2) void conv3x3_wino(const WinoArgs p) {
    constexpr int NPIXMAX = WinoCfg<MT>::NPIXMAX;
    constexpr int NI = WinoCfg<MT>::NI;
    constexpr int RAWB = (2 * NPIXMAX + 1) * 16;                   // bytes per raw buffer (+1 dump slot for idle loader lanes)

    __shared__ __attribute__((aligned(16))) char raw[2][RAWB];
    __shared__ __attribute__((aligned(16))) float ex[4 * 2 * 32 * 32];      // epilogue exchange [xi][b][tile][n]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int xi = tid >> 6;

    // ---- XCD-aware tile id (bijective) ----------------------------------------------------------------------------
    const int nblk = gridDim.x;
    const int b = blockIdx.x;
    const int xcd = b & 7;
    int tm, tn;
    if (p.nsplit > 1) {
        // A layer whose transformed weights (16 * K * C * 4 bytes) exceed an XCD's 4 MiB L2 re-streams ALL of them from the Infinity
        // Cache for every generation of resident blocks when each XCD walks every n-tile (the 8x8 -> 6x6 x 1024 trunk conv: 16.8 MB
        // of weights, 3.7 GB fetched per launch for 0.19 GB of input -- profiles/r05_pmc_traffic.json).  Split: XCD x (= blockIdx & 7)
        // streams only the n-tile subset g = x % nsplit -- its share of the weights stays L2-resident -- and the XCDs of a group share
        // partition pi = x / nsplit of the m-blocks; every input patch is then read by nsplit XCDs (patch loads run two chunks ahead
        // and tolerate the miss).  The tile -> (tm, tn) map changes, no tile's arithmetic does: results are bit-identical.
        const int g = xcd % p.nsplit, pi = xcd / p.nsplit;
        const int j = b >> 3;
        const int tml = j / p.tns;
        tn = g * p.tns + (j - tml * p.tns);
        const int size = p.part_q + (pi < p.part_r ? 1 : 0);
        if (tml >= size) return;                           // padding blocks of the shorter partitions
        tm = pi * p.part_q + min(pi, p.part_r) + tml;
    } else {
        const int q8 = nblk >> 3, rem8 = nblk & 7;
        const int tile = (xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8) + (b >> 3);
        tm = tile / p.tiles_n;
        tn = tile - tm * p.tiles_n;
    }
    const int per_img = p.per_img;
    int rb = tm - (tm / per_img) * per_img;
    int reg = 0;                                           // region of this block (wave-uniform scalar work)
    if (p.nreg > 1 && rb >= p.bx[0] * p.by[0]) {
        rb -= p.bx[0] * p.by[0];
        reg = 1;
        if (p.nreg > 2 && rb >= p.bx[1] * p.by[1]) { rb -= p.bx[1] * p.by[1]; reg = 2; }
    }
    const int TX = p.TX[reg], TY = p.TY[reg];
    const int byi = rb / p.bx[reg];
    const int bxi = rb - byi * p.bx[reg];
    const int tys = p.tiles_y, pitch = 2 * tys + 2;
    const int R0 = tm * TY;                                // stacked mode: first tile row (global) of this block
    const int n_img = p.stack ? R0 / tys : tm / per_img;   // first image of this block
    const int prow0 = p.stack ? 2 * (R0 - n_img * tys) : 0;    // stacked mode: padded row (of image n_img) the patch starts at
    const int n_here = min(p.G, p.N - n_img);              // images of this block that exist
    const int ty0 = p.stack ? 0 : p.ry0[reg] + byi * TY, tx0 = p.stack ? 0 : p.rx0[reg] + bxi * TX;   // first tile of this block
    const int tye = p.rye[reg], txe = p.rxe[reg];          // the region's end (tiles beyond belong to other blocks)
    const int iy0 = 2 * ty0 - p.pad, ix0 = 2 * tx0 - p.pad; // top-left input pixel of the raw patch (regions mode)

    const int PW = 2 * TX + 2, PH = p.stack ? p.PH : 2 * TY + 2;
    const int NPIX = PW * PH;
    const int HS = TX + 1;                                 // 16-byte entries per (patch row, x parity)
    const int nslots = TX * TY;
    // tile slot -> (image offset g, tile row ty / column tx inside the image, first patch row, validity)
    auto slot = [&](int id, int& g, int& ty, int& tx, int& prow) -> bool {
        const int r = id / TX;
        tx = tx0 + (id - r * TX);
        if (p.stack) {
            const int R = R0 + r;
            const int n = R / tys;
            g = n - n_img;
            ty = R - n * tys;
            prow = pitch * g + 2 * ty - prow0;
            return id < nslots && n < p.N;
        }
        g = 0;
        ty = ty0 + r;
        prow = 2 * r;
        return id < nslots && ty < tye && tx < txe;
    };

    // ---- raw patch loader -----------------------------------------------------------------------------------------
    const size_t img_bytes = (size_t)p.H * p.W * p.C * 4;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x + (size_t)n_img * img_bytes), 0, (int)(img_bytes * n_here), 0x00020000);
    unsigned goff[NI];
    int loff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = tid + 256 * i;
        const int half = idx & 1;
        const int pix = idx >> 1;
        const bool ok = pix < NPIX;
        const int v = pix / PW;                            // patch row
        const int px = pix - v * PW;
        int g = 0, gy = iy0 + v;
        if (p.stack) {
            const int vr = prow0 + v;
            g = vr / pitch;
            gy = vr - g * pitch - p.pad;
        }
        const int gx = ix0 + px;
        const bool inb = ok && g < n_here && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        goff[i] = inb ? (unsigned)((((g * p.H + gy) * p.W + gx) * p.C + half * 4) * 4) : kOob;
        loff[i] = ok ? (half * NPIX + (v * 2 + (px & 1)) * HS + (px >> 1)) * 16 : 2 * NPIXMAX * 16;
    }
    const int last_chunk = p.nchunks - 1;
    f32x4 rset[2][NI];
    auto load_raw = [&](f32x4 (&rs)[NI], int chunk) {
        const int c = chunk < last_chunk ? chunk : last_chunk;     // past the end: re-read the last chunk (never used)
#pragma unroll
        for (int i = 0; i < NI; ++i)
            rs[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, goff[i], c * 32, 0));
    };
    auto store_raw = [&](const f32x4 (&rs)[NI], int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(&raw[buf][loff[i]]) = rs[i];
    };

    // ---- weight fragments: [tn][chunk][p = 4*xi + nu][lane][4] ----------------------------------------------------
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.u + (size_t)tn * p.nchunks * 16384), 0, p.nchunks * 16384, 0x00020000);
    const int uoff = (xi * 4 * 64 + lane) * 16;
    auto load_b = [&](f32x4 (&bf)[4], int chunk) {
        const int c = chunk < last_chunk ? chunk : last_chunk;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
            bf[nu] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, uoff + nu * 1024, c * 16384, 0));
    };

    // ---- input transform: row xi of Bt d, then the four column combinations, in MFMA A-fragment layout -------------
    // Bt rows: xi0: d0 - d2, xi1: d1 + d2, xi2: d2 - d1, xi3: d1 - d3   =>  T = d[ra] + cb * d[rb]
    const int ra = xi == 0 ? 0 : xi == 2 ? 2 : 1;
    const int rbw = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
    const float cb = xi == 1 ? 1.f : -1.f;
    const int row_bytes = 2 * HS * 16;                      // one patch row = two parity rows
    int rbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int g, ty, tx, prow;
        if (!slot(mt * 32 + (lane & 31), g, ty, tx, prow)) slot(0, g, ty, tx, prow);    // idle slots read tile 0 (never stored)
        rbase[mt] = ((lane >> 5) * NPIX + prow * 2 * HS + (tx - tx0)) * 16;
    }
    const int oa = ra * row_bytes, ob = rbw * row_bytes;
    const int c1 = HS * 16;                                 // column offsets: j=0: 0, j=1: HS*16, j=2: 16, j=3: HS*16+16
    // The transform of one A-fragment set is cut into pieces that are pinned between individual MFMAs (sched_barrier):
    // a wave issues in order, so an s_waitcnt on LDS data right behind the ds_read would idle the matrix pipe.
    //   rd02: read columns 0, 2      c02: T0, T2, V0 = T0 - T2      rd13: read columns 1, 3
    //   c13a: T1, T3                 c13b: V1 = T1 + T2, V2 = T2 - T1, V3 = T1 - T3
    f32x4 xa0, xb0, xa2, xb2, xa1, xb1, xa3, xb3, t1, t2, t3;
    auto rd02 = [&](int buf, int mt) {
        const char* base = &raw[buf][rbase[mt]];
        xa0 = *reinterpret_cast<const f32x4*>(base + oa);
        xb0 = *reinterpret_cast<const f32x4*>(base + ob);
        xa2 = *reinterpret_cast<const f32x4*>(base + oa + 16);
        xb2 = *reinterpret_cast<const f32x4*>(base + ob + 16);
    };
    auto rd13 = [&](int buf, int mt) {
        const char* base = &raw[buf][rbase[mt]];
        xa1 = *reinterpret_cast<const f32x4*>(base + oa + c1);
        xb1 = *reinterpret_cast<const f32x4*>(base + ob + c1);
        xa3 = *reinterpret_cast<const f32x4*>(base + oa + c1 + 16);
        xb3 = *reinterpret_cast<const f32x4*>(base + ob + c1 + 16);
    };
    const f32x2 cb2 = {cb, cb};
    auto c02 = [&](f32x4 (&v)[4]) {
        const f32x4 t0 = fma4(cb2, xb0, xa0);
        t2 = fma4(cb2, xb2, xa2);
        v[0] = sub4(t0, t2);
    };
    auto c13a = [&]() {
        t1 = fma4(cb2, xb1, xa1);
        t3 = fma4(cb2, xb3, xa3);
    };
    auto c13b = [&](f32x4 (&v)[4]) {
        v[1] = add4(t1, t2);
        v[2] = sub4(t2, t1);
        v[3] = sub4(t1, t3);
    };
    auto transform = [&](f32x4 (&v)[4], int buf, int mt) {
        rd02(buf, mt); rd13(buf, mt); c02(v); c13a(); c13b(v);
    };

    f32x16 acc[4][MT];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][mt][r] = 0.f;

#define SB() __builtin_amdgcn_sched_barrier(0)
#define A1(x) do { if (!(SEAM_WINO_ABL & 1)) { x; } } while (0)
#define A8(x) do { if (!(SEAM_WINO_ABL & 8)) { x; } } while (0)
#define MF(v, bf, mt, kk, nu) acc[nu][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu][kk], bf[nu][kk], acc[nu][mt], 0, 0, 0)

    // ---- prologue -------------------------------------------------------------------------------------------------
    f32x4 bf0[4], bf1[4];
    f32x4 va[4], vb[4];
    load_raw(rset[0], 0);
    load_raw(rset[1], 1);
    load_b(bf0, 0);
    store_raw(rset[0], 0);
    store_raw(rset[1], 1);
    load_raw(rset[0], 2);
    load_raw(rset[1], 3);
    __syncthreads();
    transform(va, 0, 0);
    if (SEAM_WINO_ABL & 8) transform(vb, 1, 0);
    if (SEAM_WINO_ABL & 2) load_b(bf1, 1);
    if constexpr (MT == 1) __syncthreads();      // chunk 0 overwrites raw[0] right away

    // At the top of chunk t: raw[t&1] = patch(t), raw[(t+1)&1] = patch(t+1) (both visible), rset[t&1] = patch(t+2) in
    // flight, rset[(t+1)&1] = patch(t+3) in flight, bcur = weights(t), vcur = A fragments of (t, mt = 0).
    auto load_b2 = [&](f32x4 (&bf)[4], int chunk, int h) {      // half of load_b
        if (SEAM_WINO_ABL & 2) return;
        const int c = chunk < last_chunk ? chunk : last_chunk;
#pragma unroll
        for (int nu = 2 * h; nu < 2 * h + 2; ++nu)
            bf[nu] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, uoff + nu * 1024, c * 16384, 0));
    };
    auto chunk = [&](int t, int par, f32x4 (&bcur)[4], f32x4 (&bnext)[4], f32x4 (&vcur)[4], f32x4 (&vnext)[4]) {
        if constexpr (MT == 2) {
            // first half: MFMAs of (t, mt 0) from vcur; build vnext = fragments of (t, mt 1) from raw[par]
            SB(); MF(vcur, bcur, 0, 0, 0); A8(rd02(par, 1));
            SB(); MF(vcur, bcur, 0, 0, 1); load_b2(bnext, t + 1, 0);
            SB(); MF(vcur, bcur, 0, 0, 2); load_b2(bnext, t + 1, 1);
            SB(); MF(vcur, bcur, 0, 0, 3);
            SB(); MF(vcur, bcur, 0, 1, 0); A8(c02(vnext));
            SB(); MF(vcur, bcur, 0, 1, 1); A8(rd13(par, 1));
            SB(); MF(vcur, bcur, 0, 1, 2);
            SB(); MF(vcur, bcur, 0, 1, 3);
            SB(); MF(vcur, bcur, 0, 2, 0); A8(c13a());
            SB(); MF(vcur, bcur, 0, 2, 1); A8(c13b(vnext));
            SB(); MF(vcur, bcur, 0, 2, 2);
            SB(); MF(vcur, bcur, 0, 2, 3);
            SB(); MF(vcur, bcur, 0, 3, 0);
            SB(); MF(vcur, bcur, 0, 3, 1);
            SB(); MF(vcur, bcur, 0, 3, 2);
            SB(); MF(vcur, bcur, 0, 3, 3);
            SB();
            if (!(SEAM_WINO_ABL & 4)) __syncthreads();                       // every wave is done reading raw[par]
            // second half: MFMAs of (t, mt 1) from vnext; patch(t+2) -> raw[par]; vcur = fragments of (t+1, mt 0)
            SB(); MF(vnext, bcur, 1, 0, 0); A1(store_raw(rset[par], par));
            SB(); MF(vnext, bcur, 1, 0, 1); A8(rd02(par ^ 1, 0));
            SB(); MF(vnext, bcur, 1, 0, 2); A1(load_raw(rset[par], t + 4));
            SB(); MF(vnext, bcur, 1, 0, 3);
            SB(); MF(vnext, bcur, 1, 1, 0); A8(c02(vcur));
            SB(); MF(vnext, bcur, 1, 1, 1); A8(rd13(par ^ 1, 0));
            SB(); MF(vnext, bcur, 1, 1, 2);
            SB(); MF(vnext, bcur, 1, 1, 3);
            SB(); MF(vnext, bcur, 1, 2, 0); A8(c13a());
            SB(); MF(vnext, bcur, 1, 2, 1); A8(c13b(vcur));
            SB(); MF(vnext, bcur, 1, 2, 2);
            SB(); MF(vnext, bcur, 1, 2, 3);
            SB(); MF(vnext, bcur, 1, 3, 0);
            SB(); MF(vnext, bcur, 1, 3, 1);
            SB(); MF(vnext, bcur, 1, 3, 2);
            SB(); MF(vnext, bcur, 1, 3, 3);
            SB();
        } else {
            // one step per chunk: raw[par] (patch t) was consumed during chunk t-1; vnext = fragments of chunk t+1
            SB(); MF(vcur, bcur, 0, 0, 0); A1(store_raw(rset[par], par));
            SB(); MF(vcur, bcur, 0, 0, 1); A8(rd02(par ^ 1, 0));
            SB(); MF(vcur, bcur, 0, 0, 2); A1(load_raw(rset[par], t + 4));
            SB(); MF(vcur, bcur, 0, 0, 3); load_b2(bnext, t + 1, 0);
            SB(); MF(vcur, bcur, 0, 1, 0); A8(c02(vnext));
            SB(); MF(vcur, bcur, 0, 1, 1); A8(rd13(par ^ 1, 0));
            SB(); MF(vcur, bcur, 0, 1, 2); load_b2(bnext, t + 1, 1);
            SB(); MF(vcur, bcur, 0, 1, 3);
            SB(); MF(vcur, bcur, 0, 2, 0); A8(c13a());
            SB(); MF(vcur, bcur, 0, 2, 1); A8(c13b(vnext));
            SB(); MF(vcur, bcur, 0, 2, 2);
            SB(); MF(vcur, bcur, 0, 2, 3);
            SB(); MF(vcur, bcur, 0, 3, 0);
            SB(); MF(vcur, bcur, 0, 3, 1);
            SB(); MF(vcur, bcur, 0, 3, 2);
            SB(); MF(vcur, bcur, 0, 3, 3);
            SB();
            __syncthreads();
        }
    };
    for (int t = 0; t < p.nchunks; t += 2) {
        if constexpr (MT == 2) {
            chunk(t, 0, bf0, bf1, va, vb);
            if (t + 1 < p.nchunks) chunk(t + 1, 1, bf1, bf0, va, vb);
        } else {
            chunk(t, 0, bf0, bf1, va, vb);
            if (t + 1 < p.nchunks) chunk(t + 1, 1, bf1, bf0, vb, va);
        }
    }
#undef SB
#undef A1
#undef A8
#undef MF

    // ---- epilogue: output transform + scale/shift (+ residual, ReLU) ------------------------------------------------
    const size_t out_img = (size_t)p.Ho * p.Wo * p.K * 4;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)p.y + (size_t)n_img * out_img), 0, (int)(out_img * n_here), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(p.res ? p.res : p.y) + (size_t)n_img * out_img), 0, (int)(out_img * n_here), 0x00020000);
    const int et = tid >> 3;          // tile row of the exchange this thread finishes
    const int n4 = tid & 7;
    const int ncol = tn * 32 + n4 * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + ncol);
    if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + ncol);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        __syncthreads();              // previous readers of `ex` (and, first time, of `raw`) are done
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float m0 = acc[0][mt][r], m1 = acc[1][mt][r], m2 = acc[2][mt][r], m3 = acc[3][mt][r];
            ex[((xi * 2 + 0) * 32 + row) * 32 + (lane & 31)] = m0 + m1 + m2;
            ex[((xi * 2 + 1) * 32 + row) * 32 + (lane & 31)] = m1 - m2 - m3;
        }
        __syncthreads();
        int g, tyt, txt, prow_unused;
        const bool tile_ok = slot(mt * 32 + et, g, tyt, txt, prow_unused) && g < n_here;
        const int oy = 2 * tyt, ox = 2 * txt;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(&ex[((0 * 2 + bb) * 32 + et) * 32 + n4 * 4]);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(&ex[((1 * 2 + bb) * 32 + et) * 32 + n4 * 4]);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(&ex[((2 * 2 + bb) * 32 + et) * 32 + n4 * 4]);
            const f32x4 q3 = *reinterpret_cast<const f32x4*>(&ex[((3 * 2 + bb) * 32 + et) * 32 + n4 * 4]);
            f32x4 yv[2];
            yv[0] = q0 + q1 + q2;
            yv[1] = q1 - q2 - q3;
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                const bool ok = tile_ok && (oy + aa) < p.Ho && (ox + bb) < p.Wo;
                const unsigned off = ok ? (unsigned)(((((g * p.Ho + oy + aa) * p.Wo) + ox + bb) * p.K + ncol) * 4) : kOob;
                f32x4 v = yv[aa] * sc + sh;
                if (p.res) {
                    const f32x4 rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0));
                    if (p.relu == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = rv[e] > 0.f ? v[e] : 0.f;
                    } else {
                        v += rv;
                    }
                }
                if (p.relu == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc, off, 0, 0);
            }
        }
    }
}

// OIHW fp32 [K, Cin, 3, 3] -> U = G g Gt in MFMA fragment order: [K/32][Cstore/8][16][64][4]
//   element (tn, chunk, p = 4*xi + nu, lane, e) = U_p[n = 32*tn + (lane & 31)][c = 8*chunk + 4*(lane >> 5) + e]
// mode 0: forward weights; mode 2: input-gradient weights of a Conv2d [Cin(out) , K(in), 3, 3] (taps rotated 180 degrees,
// channels swapped), as in seam_pack_conv_weight_f32.
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int K, int Cin, int Cstore, int mode) {
    const int nch = Cstore / 8;
    const size_t total = (size_t)(K / 32) * nch * 64;      // one thread per (tn, chunk, lane): 4 channels x 16 positions
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t rest = i >> 6;
        const int chunk = (int)(rest % nch);
        const int tn = (int)(rest / nch);
        const int n = tn * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = chunk * 8 + (lane >> 5) * 4 + e;
            double g[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    float v = 0.f;
                    if (c < Cin) v = mode == 2 ? w[(((size_t)c * K + n) * 3 + (2 - r)) * 3 + (2 - s)] : w[(((size_t)n * Cin + c) * 3 + r) * 3 + s];
                    g[r][s] = (double)v;
                }
            double gg[4][3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                gg[0][s] = g[0][s];
                gg[1][s] = 0.5 * (g[0][s] + g[1][s] + g[2][s]);
                gg[2][s] = 0.5 * (g[0][s] - g[1][s] + g[2][s]);
                gg[3][s] = g[2][s];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double u0 = gg[q][0];
                const double u1 = 0.5 * (gg[q][0] + gg[q][1] + gg[q][2]);
                const double u2 = 0.5 * (gg[q][0] - gg[q][1] + gg[q][2]);
                const double u3 = gg[q][2];
                const size_t base = ((((size_t)tn * nch + chunk) * 16 + q * 4) * 64 + lane) * 4 + e;
                out[base] = (float)u0;
                out[base + 256] = (float)u1;
                out[base + 512] = (float)u2;
                out[base + 768] = (float)u3;
            }
        }
    }
}

// Block layout for one tile variant (32*MT tile slots per block).  Small maps: the whole tile grid of G images per block.
// Otherwise up to three regions per image: a main region tiled exactly by TX x TY patches, and the right / bottom strips
// that remain, each tiled by the best patch for ITS shape -- a 100 x 100 tile map takes 157 blocks of 64 instead of 169.
struct Layout {
    int nreg, G, stack, PH;
    int rx0[3], ry0[3], rxe[3], rye[3], TX[3], TY[3], bx[3], by[3];
    long per_img, blocks;      // blocks: per n-tile, for N images
};

inline bool patch_ok(int tx, int ty, int cap, int npixmax) { return tx >= 1 && ty >= 1 && tx * ty <= cap && (2 * tx + 2) * (2 * ty + 2) <= npixmax; }

// best uniform tiling of a w x h tile rectangle: fewest blocks, then the smallest raw patch
inline long best_uniform(int w, int h, int cap, int npixmax, int& TX, int& TY) {
    long best = -1, best_cost = -1;
    for (int ty = 1; ty <= cap && ty <= h; ++ty)
        for (int tx = 1; tx * ty <= cap && tx <= w; ++tx) {
            if (!patch_ok(tx, ty, cap, npixmax)) continue;
            const long nb = (long)((w + tx - 1) / tx) * ((h + ty - 1) / ty);
            const long cost = nb * 4096 + (2 * tx + 2) * (2 * ty + 2);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nb; TX = tx; TY = ty; }
        }
    return best;
}

inline Layout choose_layout(int N, int tiles_x, int tiles_y, int mt, size_t in_img_bytes, size_t out_img_bytes) {
    const int cap = 32 * mt, npixmax = mt == 2 ? 384 : 208;
    Layout L;
    L.nreg = 1; L.G = 1; L.stack = 0; L.PH = 0;
    // Stacked candidate (maps narrower than a block): TY consecutive tile rows of the batch at full width per block.
    Layout S;
    S.blocks = -1;
    if (tiles_x <= cap) {
        const int t = tiles_y, pitch = 2 * t + 2, pw = 2 * tiles_x + 2;
        for (int ty = cap / tiles_x; ty >= 1; --ty) {
            int phmax = 0, gmax = 0;
            for (int b = 0; b < t; ++b) {                                // block starts repeat with period <= tiles_y
                const int tyf = (int)(((long)b * ty) % t);
                const int last = tyf + ty - 1;
                const int gl = last / t, tyl = last - gl * t;
                const int span = pitch * gl + 2 * tyl + 4 - 2 * tyf;
                if (span > phmax) phmax = span;
                if (gl + 1 > gmax) gmax = gl + 1;
            }
            if (phmax * pw > npixmax) continue;
            if ((size_t)gmax * in_img_bytes >= kOob || (size_t)gmax * out_img_bytes >= kOob) continue;
            S.nreg = 1; S.stack = 1; S.PH = phmax; S.G = gmax;
            S.rx0[0] = S.ry0[0] = 0; S.rxe[0] = tiles_x; S.rye[0] = tiles_y; S.TX[0] = tiles_x; S.TY[0] = ty; S.bx[0] = S.by[0] = 1;
            S.per_img = 1;
            S.blocks = ((long)N * t + ty - 1) / ty;
            break;
        }
    }
    // one region (uniform tiling) is the baseline
    long best_blocks = best_uniform(tiles_x, tiles_y, cap, npixmax, L.TX[0], L.TY[0]);
    L.rx0[0] = L.ry0[0] = 0; L.rxe[0] = tiles_x; L.rye[0] = tiles_y;
    L.bx[0] = (tiles_x + L.TX[0] - 1) / L.TX[0]; L.by[0] = (tiles_y + L.TY[0] - 1) / L.TY[0];
    // main region + strips
    for (int ty = 1; ty <= cap && ty <= tiles_y; ++ty)
        for (int tx = 1; tx * ty <= cap && tx <= tiles_x; ++tx) {
            if (!patch_ok(tx, ty, cap, npixmax)) continue;
            const int mx = tiles_x / tx, my = tiles_y / ty;              // exact blocks of the main region
            const int wm = mx * tx, hm = my * ty;
            if (wm == 0 || hm == 0) continue;
            Layout C;
            C.G = 1; C.nreg = 1; C.stack = 0; C.PH = 0;
            C.rx0[0] = 0; C.ry0[0] = 0; C.rxe[0] = wm; C.rye[0] = hm; C.TX[0] = tx; C.TY[0] = ty; C.bx[0] = mx; C.by[0] = my;
            long nb = (long)mx * my;
            if (wm < tiles_x) {                                          // right strip: full height
                const int r = C.nreg++;
                const long b = best_uniform(tiles_x - wm, tiles_y, cap, npixmax, C.TX[r], C.TY[r]);
                C.rx0[r] = wm; C.ry0[r] = 0; C.rxe[r] = tiles_x; C.rye[r] = tiles_y;
                C.bx[r] = (tiles_x - wm + C.TX[r] - 1) / C.TX[r]; C.by[r] = (tiles_y + C.TY[r] - 1) / C.TY[r];
                nb += b;
            }
            if (hm < tiles_y) {                                          // bottom strip: under the main region only
                const int r = C.nreg++;
                const long b = best_uniform(wm, tiles_y - hm, cap, npixmax, C.TX[r], C.TY[r]);
                C.rx0[r] = 0; C.ry0[r] = hm; C.rxe[r] = wm; C.rye[r] = tiles_y;
                C.bx[r] = (wm + C.TX[r] - 1) / C.TX[r]; C.by[r] = (tiles_y - hm + C.TY[r] - 1) / C.TY[r];
                nb += b;
            }
            if (nb < best_blocks) {
                best_blocks = nb;
                L = C;
            }
        }
    L.per_img = 0;
    for (int r = 0; r < L.nreg; ++r) L.per_img += (long)L.bx[r] * L.by[r];
    L.blocks = L.per_img * N;
    if (S.blocks > 0 && S.blocks < L.blocks) return S;
    return L;
}

inline bool wino_ok(int C, int K, int R, int S, int stride) { return R == 3 && S == 3 && stride == 1 && C % 8 == 0 && K % 32 == 0 && C >= 8; }

}  // namespace

extern "C" {

int seam_wino_supported(int C, int K, int R, int S, int stride) { return wino_ok(C, K, R, S, stride) ? 1 : 0; }

long long seam_wino_weight_floats(int K, int Cstore) { return (long long)K * Cstore * 16; }

int seam_pack_conv_weight_wino_f32(const float* w, float* u_packed, int K, int Cin, int Cstore, int mode, void* stream) {
    if (!wino_ok(Cstore, K, 3, 3, 1) || Cin > Cstore) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)(K / 32) * (Cstore / 8) * 64;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, u_packed, K, Cin, Cstore, mode);
    return (int)hipGetLastError();
}

// Launch plan shared by the launcher and the profitability query.
static int wino_plan(WinoArgs& a, int N, int H, int W, int C, int K, int pad, int& mt, long& blocks) {
    if (!wino_ok(C, K, 3, 3, 1) || N <= 0) return (int)hipErrorInvalidValue;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K;
    a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2; a.pad = pad;
    if (a.Ho <= 0 || a.Wo <= 0) return (int)hipErrorInvalidValue;
    if ((size_t)H * W * C * 4 >= kOob || (size_t)a.Ho * a.Wo * K * 4 >= kOob) return (int)hipErrorInvalidValue;
    const int tiles_x = (a.Wo + 1) / 2, tiles_y = (a.Ho + 1) / 2;
    const int force_mt = seam_opt::get(seam_opt::WINO_MT);    // kernel experiments
    const size_t in_b = (size_t)H * W * C * 4, out_b = (size_t)a.Ho * a.Wo * K * 4;
    const Layout p2 = choose_layout(N, tiles_x, tiles_y, 2, in_b, out_b), p1 = choose_layout(N, tiles_x, tiles_y, 1, in_b, out_b);
    a.tiles_n = K / 32;
    a.nchunks = C / 8;
    // Tile variant: a 64-tile block costs ~1.9x a 32-tile block (same weight stream, twice the MFMAs); take the 32-tile
    // variant when it wastes fewer slots, or when the 64-tile grid could not fill the chip twice.
    mt = (p1.blocks * 100 < p2.blocks * 190 || p2.blocks * a.tiles_n < 1024) ? 1 : 2;
    if (force_mt == 1 || force_mt == 2) mt = force_mt;
    const Layout& pp = mt == 2 ? p2 : p1;
    a.nreg = pp.nreg; a.G = pp.G; a.per_img = (int)pp.per_img;
    a.stack = pp.stack; a.PH = pp.PH; a.tiles_y = tiles_y;
    for (int r = 0; r < 3; ++r) {
        const int q = r < pp.nreg ? r : 0;
        a.rx0[r] = pp.rx0[q]; a.ry0[r] = pp.ry0[q]; a.rxe[r] = pp.rxe[q]; a.rye[r] = pp.rye[q];
        a.TX[r] = pp.TX[q]; a.TY[r] = pp.TY[q]; a.bx[r] = pp.bx[q]; a.by[r] = pp.by[q];
    }
    blocks = pp.blocks * a.tiles_n;
    {
        // n-tile split (see the kernel's tile decode): as many XCD groups as it takes for a group's share of the weights to sit in
        // its L2 beside the patches (<= 2.5 MB), when the launch is large enough for every group to keep its CUs busy
        const int want = seam_opt::get(seam_opt::WINO_NSPLIT);      // 0: this rule; > 0: forced (experiments)
        int ns = 1;
        const size_t wbytes = (size_t)16 * K * C * 4;
        if (want > 0) ns = want;
        else while (ns < 8 && wbytes / ns > (size_t)2560 * 1024) ns <<= 1;
        while (ns > 1 && (a.tiles_n % ns || 8 % ns || pp.blocks * (a.tiles_n / ns) < 8L * 64)) ns >>= 1;
        a.nsplit = ns < 1 ? 1 : ns;
        a.tns = a.tiles_n / a.nsplit;
        const int parts = 8 / a.nsplit;
        a.part_q = (int)(pp.blocks / parts); a.part_r = (int)(pp.blocks % parts);
        if (a.nsplit > 1) blocks = 8L * (a.part_q + (a.part_r ? 1 : 0)) * a.tns;
    }
    if (blocks > 0x7fffffffL) return (int)hipErrorInvalidValue;
    return 0;
}

/* Percentage of the block's tile slots that hold real output tiles (100 = no waste).  The Winograd kernel issues
 * 2.25x fewer MFMAs than the implicit GEMM, so it wins when this is above ~50 (ops.conv2d uses it to pick). */
int seam_wino_slot_fill_pct(int N, int H, int W, int C, int K, int pad) {
    WinoArgs a;
    int mt;
    long blocks;
    if (wino_plan(a, N, H, W, C, K, pad, mt, blocks)) return 0;
    const double tiles = (double)N * ((a.Wo + 1) / 2) * ((a.Ho + 1) / 2) * a.tiles_n;
    const long work = a.nsplit > 1 ? ((long)(8 / a.nsplit) * a.part_q + a.part_r) * a.tiles_n : blocks;     // without the padding blocks
    return (int)(100.0 * tiles / ((double)work * 32 * mt));
}

/* MFMA issues of the launch in units of 32x32x8-channel position GEMMs (blocks x 32*MT tile slots x 16 positions; 0 =
 * unsupported) -- compared with seam_wino24_issue_slots to pick the cheaper Winograd form per layer shape. */
long long seam_wino_issue_slots(int N, int H, int W, int C, int K, int pad) {
    WinoArgs a;
    int mt;
    long blocks;
    if (wino_plan(a, N, H, W, C, K, pad, mt, blocks)) return 0;
    const long work = a.nsplit > 1 ? ((long)(8 / a.nsplit) * a.part_q + a.part_r) * a.tiles_n : blocks;
    return (long long)work * 32 * mt * 16;
}

int seam_wino_tile_variant(int N, int H, int W, int C, int K, int pad) {     // MT of conv3x3_wino<MT> the launcher picks (0: unsupported)
    WinoArgs a;
    int mt;
    long blocks;
    return wino_plan(a, N, H, W, C, K, pad, mt, blocks) ? 0 : mt;
}

int seam_conv3x3_wino_f32(const float* x, const float* u_packed, const float* scale, const float* shift, const float* residual,
                          float* y, int N, int H, int W, int C, int K, int pad, int relu, void* stream) {
    WinoArgs a;
    int mt;
    long blocks;
    const int rc = wino_plan(a, N, H, W, C, K, pad, mt, blocks);
    if (rc) return rc;
    a.x = x; a.u = u_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    if (mt == 2) hipLaunchKernelGGL((conv3x3_wino<2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_wino<1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

}  // extern "C"
