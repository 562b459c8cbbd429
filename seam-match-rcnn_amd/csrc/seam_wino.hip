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

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kOob = 0x80000000u;

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
    int TX, TY;           // tiles per block patch (TX*TY <= 32*MT)
    int bx, by;           // blocks per image along x / y
    int G;                // images per block (small maps: the whole tile grid of G images shares one block; bx = by = 1)
    int tiles_n;          // K / 32
    int nchunks;          // C / 8
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
    const int q8 = nblk >> 3, rem8 = nblk & 7;
    const int tile = (xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8) + (b >> 3);
    const int tm = tile / p.tiles_n;
    const int tn = tile - tm * p.tiles_n;
    const int per_img = p.bx * p.by;
    const int n_img = (tm / per_img) * p.G;                // first image of this block
    const int rb = tm - (tm / per_img) * per_img;
    const int n_here = min(p.G, p.N - n_img);              // images of this block that exist
    const int byi = rb / p.bx;
    const int bxi = rb - byi * p.bx;
    const int ty0 = byi * p.TY, tx0 = bxi * p.TX;          // first tile of this block
    const int iy0 = 2 * ty0 - p.pad, ix0 = 2 * tx0 - p.pad; // top-left input pixel of the raw patch

    const int PW = 2 * p.TX + 2, PH = 2 * p.TY + 2;
    const int NPIX1 = PW * PH;                             // patch pixels per image
    const int NPIX = NPIX1 * p.G;
    const int tpi = p.TX * p.TY;                           // tile slots per image
    const int HS = p.TX + 1;                               // 16-byte entries per (patch row, x parity)

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
        const int g = pix / NPIX1;
        const int lp = pix - g * NPIX1;
        const int py = lp / PW;
        const int px = lp - py * PW;
        const int gy = iy0 + py, gx = ix0 + px;
        const bool inb = ok && g < n_here && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        goff[i] = inb ? (unsigned)((((g * p.H + gy) * p.W + gx) * p.C + half * 4) * 4) : kOob;
        loff[i] = ok ? (half * NPIX + g * NPIX1 + (py * 2 + (px & 1)) * HS + (px >> 1)) * 16 : 2 * NPIXMAX * 16;
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
        int id = mt * 32 + (lane & 31);
        if (id >= tpi * p.G) id = 0;                       // idle tile slots read tile 0 (results are never stored)
        const int g = id / tpi;
        const int li = id - g * tpi;
        const int tyl = li / p.TX;
        const int txl = li - tyl * p.TX;
        rbase[mt] = ((lane >> 5) * NPIX + g * NPIX1 + 4 * tyl * HS + txl) * 16;
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
    auto c02 = [&](f32x4 (&v)[4]) {
        const f32x4 t0 = xa0 + cb * xb0;
        t2 = xa2 + cb * xb2;
        v[0] = t0 - t2;
    };
    auto c13a = [&]() {
        t1 = xa1 + cb * xb1;
        t3 = xa3 + cb * xb3;
    };
    auto c13b = [&](f32x4 (&v)[4]) {
        v[1] = t1 + t2;
        v[2] = t2 - t1;
        v[3] = t1 - t3;
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
        const int id = mt * 32 + et;
        const int g = id / tpi;
        const int li = id - g * tpi;
        const bool tile_ok = g < n_here;
        const int tyl = li / p.TX;
        const int txl = li - tyl * p.TX;
        const int oy = 2 * (ty0 + tyl), ox = 2 * (tx0 + txl);
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

// Patch shape for one tile variant: minimise the number of blocks (then the raw patch size) over TX*TY <= 32*MT; maps
// whose whole tile grid fits several times into a block share it between G images (ROI tiles of the heads).
struct Patch { int TX, TY, G, bx, by; long blocks; };   // blocks: per n-tile, for N images

inline Patch choose_patch(int N, int tiles_x, int tiles_y, int mt, size_t in_img_bytes, size_t out_img_bytes) {
    const int cap = 32 * mt, npixmax = mt == 2 ? 384 : 208;
    Patch best;
    best.blocks = -1;
    const int whole = (2 * tiles_x + 2) * (2 * tiles_y + 2);
    if (tiles_x * tiles_y <= cap && whole <= npixmax) {          // whole images per block
        int g = cap / (tiles_x * tiles_y);
        if (g * whole > npixmax) g = npixmax / whole;
        while (g > 1 && ((size_t)g * in_img_bytes >= kOob || (size_t)g * out_img_bytes >= kOob)) --g;
        if (g > N) g = N;
        best.TX = tiles_x; best.TY = tiles_y; best.G = g; best.bx = best.by = 1;
        best.blocks = (N + g - 1) / g;
        return best;
    }
    long best_cost = -1;
    for (int ty = 1; ty <= cap; ++ty)
        for (int tx = 1; tx * ty <= cap; ++tx) {
            const int npix = (2 * tx + 2) * (2 * ty + 2);
            if (npix > npixmax) continue;
            const long nb = (long)((tiles_x + tx - 1) / tx) * ((tiles_y + ty - 1) / ty);
            const long cost = nb * 4096 + npix;
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best.TX = tx; best.TY = ty; best.G = 1;
                best.bx = (tiles_x + tx - 1) / tx; best.by = (tiles_y + ty - 1) / ty;
                best.blocks = nb * N;
            }
        }
    return best;
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
    static const int force_mt = getenv("SEAM_WINO_MT") ? atoi(getenv("SEAM_WINO_MT")) : 0;    // kernel experiments
    const size_t in_b = (size_t)H * W * C * 4, out_b = (size_t)a.Ho * a.Wo * K * 4;
    const Patch p2 = choose_patch(N, tiles_x, tiles_y, 2, in_b, out_b), p1 = choose_patch(N, tiles_x, tiles_y, 1, in_b, out_b);
    a.tiles_n = K / 32;
    a.nchunks = C / 8;
    // Tile variant: a 64-tile block costs ~1.9x a 32-tile block (same weight stream, twice the MFMAs); take the 32-tile
    // variant when it wastes fewer slots, or when the 64-tile grid could not fill the chip twice.
    mt = (p1.blocks * 100 < p2.blocks * 190 || p2.blocks * a.tiles_n < 1024) ? 1 : 2;
    if (force_mt == 1 || force_mt == 2) mt = force_mt;
    const Patch& pp = mt == 2 ? p2 : p1;
    a.TX = pp.TX; a.TY = pp.TY; a.G = pp.G; a.bx = pp.bx; a.by = pp.by;
    blocks = pp.blocks * a.tiles_n;
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
    return (int)(100.0 * tiles / ((double)blocks * 32 * mt));
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
