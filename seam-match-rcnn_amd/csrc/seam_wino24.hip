// seam_wino24.hip -- Winograd F(2x4,3x3) convolution on the gfx950 fp32 matrix cores.
//
// The second Winograd kernel of the path: output tiles of 2 rows x 4 columns from 4 x 6 input tiles, i.e. F(2,3) down
// the rows (as seam_wino.hip) and F(4,3) along the columns: 24 multiplies per 8 outputs = 3 per output instead of
// 4 (F(2x2,3x3)) or 9 (direct) -- 3x fewer MFMA issues than the implicit GEMM, 1.33x fewer than seam_wino.hip.
//   Y = A2t [ (G2 g G4t) (.) (B2t d B4) ] A4   per tile, summed over input channels = 24 independent GEMMs
//   M_p[tile, n] = sum_c V_p[tile, c] * U_p[n, c],  p = (xi in 0..3, nu in 0..5), all fp32 (v_mfma_f32_32x32x2_f32).
// The F(4,3) half uses the points {0, +-1, +-2, inf}; its transforms multiply by 2, 4, 5, 8 (input / output side) and
// 1/4, 1/6, 1/12, 1/24 (weights, computed in fp64 at pack time and rounded once): measured rounding ~2x that of
// F(2x2,3x3), ~1e-6 of the output scale (tools/wino_bench.py).
//
// Mapping (same ideas as seam_wino.hip -- no operand goes through LDS):
//   block = 4 waves; wave xi owns the six positions (xi, nu = 0..5) for 32 tiles x 32 output channels: 6 accumulator
//   tiles of 32x32 (96 VGPRs), two blocks per CU.
//   A operand: each wave computes ITS row of B2t d (two input rows per column) and the six column combinations in
//     registers, directly in MFMA A-fragment layout, from the block's raw input patch -- the only LDS resident:
//     (2*TY+2) x (4*TX+2) pixels x 8 channels per chunk, split by channel half and by x mod 4 so that the tile-strided
//     ds_read_b128s are conflict free, double buffered, one barrier per 8-channel chunk.
//   B operand: U packed in fragment order [n_tile][chunk][p][lane][4]; a wave streams its 6 KiB per chunk with coalesced
//     buffer loads straight into registers.
//   Registers are the scarce resource (96 accumulators): A and B fragments are SINGLE sets that roll -- the positions are
//     walked in pairs (0,5), (1,2), (3,4) (the pairs that share sub-expressions of B4t); as soon as the 8 MFMAs of a
//     pair are issued, its A fragments are overwritten with those of the next chunk and its weight loads for the next
//     chunk are issued (16 MFMA slots ahead of their use).
//   Epilogue: the nu half of the output transform (A4t: 6 -> 4) in registers, the xi half (A2t: 4 -> 2) through a
//     32 KiB LDS exchange in two passes, scale / shift (+ residual, ReLU), 16-byte NHWC stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "seam_opts.h"
#if defined(SEAM_W24PC_TRACE)
#include "dev/seam_trace_host.h"      // -DSEAM_DEV_BUILD experiment builds only (tools/experiments/w24pc_abl.sh)
#endif
#include <atomic>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned kOob = 0x80000000u;

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
#ifndef SEAM_W24_PK
#define SEAM_W24_PK 1       // 1 (default): the in-loop input transform on packed fp32 VALU ops; 0: scalar-lane v_fma_f32 / v_add_f32
#endif
// The input transform's VALU work sits between fp32 MFMAs.  MI355X_MICROARCH.md lists packed fp32 VALU instructions as an
// anti-lever beside (bf16) MFMAs; beside FP32 MFMAs that does not hold -- measured on the bench's twelve layer shapes
// (profiles/r02_w24_scalar_valu.txt, one box): the scalar-lane form (72 v_fma/add/sub per chunk, SEAM_W24_PK=0) is 1-2 % SLOWER
// than the packed form (36 v_pk_*): the fp32 MFMA and the fp32 VALU share the SIMD's FMA lanes, what counts is FMAs, not opcodes.
__device__ __forceinline__ float s_fma(float c, float b, float a) {        // a + c * b, c wave-uniform (SGPR)
    float d;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(c), "v"(b), "v"(a));
    return d;
}
__device__ __forceinline__ float s_add(float a, float b) {
    float d;
    asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float s_sub(float a, float b) {
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x4 fma4(f32x2 c, f32x4 b, f32x4 a) {        // a + c * b
    const f32x2 lo = pk_fma(c, __builtin_shufflevector(b, b, 0, 1), __builtin_shufflevector(a, a, 0, 1));
    const f32x2 hi = pk_fma(c, __builtin_shufflevector(b, b, 2, 3), __builtin_shufflevector(a, a, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// the same with a wave-uniform multiplier held in an SGPR pair (one scalar source per VOP3P): the constants of B4t cost no
// VGPRs -- the register file is full (96 accumulators, two weight sets), and a VGPR temporary that aliases the
// destination of an in-flight weight load makes hipcc wait for that load
__device__ __forceinline__ f32x2 pk_fma_s(f32x2 cs, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(cs), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x4 fma4s(f32x2 cs, f32x4 b, f32x4 a) {      // a + cs * b
#if SEAM_W24_PK
    const f32x2 lo = pk_fma_s(cs, __builtin_shufflevector(b, b, 0, 1), __builtin_shufflevector(a, a, 0, 1));
    const f32x2 hi = pk_fma_s(cs, __builtin_shufflevector(b, b, 2, 3), __builtin_shufflevector(a, a, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
#else
    const float c = cs[0];
    return f32x4{s_fma(c, b[0], a[0]), s_fma(c, b[1], a[1]), s_fma(c, b[2], a[2]), s_fma(c, b[3], a[3])};
#endif
}
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) {
#if SEAM_W24_PK
    const f32x2 lo = pk_add(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_add(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
#else
    return f32x4{s_add(a[0], b[0]), s_add(a[1], b[1]), s_add(a[2], b[2]), s_add(a[3], b[3])};
#endif
}
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
#if SEAM_W24_PK
    const f32x2 lo = pk_sub(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_sub(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
#else
    return f32x4{s_sub(a[0], b[0]), s_sub(a[1], b[1]), s_sub(a[2], b[2]), s_sub(a[3], b[3])};
#endif
}

#ifndef SEAM_W24_ABL
#define SEAM_W24_ABL 0      // kernel experiments (tools/experiments/wino24_abl.sh; operands keep the REAL data of chunks 0/1):
                            // 1 no in-loop patch loads / LDS stores, 2 no in-loop weight loads, 4 no barrier, 8 no in-loop transforms, 16 no epilogue
#endif

struct Wino24Args {
    const float* x;
    const float* u;       // packed transformed weights
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int N, H, W, C, K;
    int Ho, Wo, pad, relu;
    // Up to three regions tile an image (main region + right strip + bottom strip), each with its own block patch shape,
    // or -- stacked mode, maps narrower than a block -- TY[0] consecutive tile rows of the whole batch per block
    // (see seam_wino.hip; identical conventions, only the tile is 4 output columns wide).
    int nreg;
    int rx0[3], ry0[3];
    int rxe[3], rye[3];
    int TX[3], TY[3];     // tiles per block patch (TX*TY <= 32)
    int bx[3], by[3];
    int per_img;
    int stack;
    int tiles_y;
    int PH;
    int G;
    int nt;               // n-tiles per block (kernel template NT)
    int tiles_n;          // K / (32 * NT)
    int nchunks;          // C / 8
    // ceil(2^32 / d) of the divisors the block prologue needs (fdiv below): an integer division costs ~20 VALU instructions,
    // and VALU instructions of either resident block delay the matrix pipe
    unsigned m_tiles_n, m_per_img, m_tys, m_pitch, m_bx[3], m_TX[3], m_PW[3];
    // n-tile split over XCD groups (see the kernel's tile decode): nsplit groups, tns = tiles_n / nsplit n-tiles per group, the
    // patches (tm) cut into 8 / nsplit partitions of part_q (+1 for the first part_r) each
    int nsplit, tns, part_q, part_r;
    unsigned m_tns;
    int total_tiles;             // conv3x3_wino24pc: tiles of the launch (the grid is persistent: min(tiles, CUs) blocks)
    // conv3x3_wino24pc: a region's parameters side by side -- ONE wide scalar load per tile decode instead of a dozen dependent ones
    // (the kernel re-derives a tile's geometry from its index in several places; every dependent s_load is ~250 cycles of latency)
    struct Rg { int TX, TY, bx, by, rx0, ry0, rxe, rye; unsigned m_bx, m_TX, m_PW; int pad_[5]; } rg[3];
    unsigned long long* trace;   // SEAM_W24PC_TRACE builds only: s_memtime stamps of one block's waves 0 and 4 (else null)
};

// a / d for 0 <= a, a * d < 2^32, with m = ceil(2^32 / d) (d >= 2) -- one v_mul_hi_u32 / s_mul_hi_u32
__device__ __forceinline__ int fdiv(int a, int d, unsigned m) { return d == 1 ? a : (int)__umulhi((unsigned)a, m); }

constexpr int NPIXMAX = 384;                       // raw patch pixels per buffer (3 x 16-byte loads per thread per chunk)
constexpr int NI = (2 * NPIXMAX + 255) / 256;
constexpr int ENTMAX = 672;                        // 16-byte LDS entries per channel half (row pairs x PR, see the kernel)
constexpr int RAWB = (2 * ENTMAX + 1) * 16;        // bytes per raw buffer (+1 dump slot for idle loader lanes)

// NT = 32-channel n-tiles per block.  NT = 1: 96 accumulator VGPRs, two blocks per CU.  NT = 2: 192 accumulators (AccVGPRs), one block
// per CU, one wave per SIMD: every A fragment (the transform's output) and the raw patch feed twice the MFMAs.  Measured on this
// kernel: a SIMD does not overlap its MFMAs with anything else it issues -- one resident block instead of two costs only 10 %,
// MFMAs-only runs at the same speed with one or two blocks, and the K loop's time is the sum of its MFMA cycles and ~16 cycles per
// other vector instruction -- so what counts is MFMAs per transform / LDS / load instruction, not occupancy.
// HSC: the LDS phase stride HS = (TX + 1) | 1 as a compile-time constant (0 = take it from TX at run time).  With a constant HS every
// ds_read_b128 of the input transform is one base register + an immediate offset; with a run-time HS the compiler keeps 8-12 loop-
// invariant address registers per buffer alive through the K loop (and, in the NT = 2 form, spills them to AccVGPRs and reads them
// back every chunk).  The kernel dispatches on the block's region: HS = 9 (TX 7..8: the main region of every large map), 5 (TX 3..4),
// 3 (TX <= 2: the right-hand strips), anything else through the run-time form.
template <int NT, int HSC>
__device__ __forceinline__ void w24_block(const Wino24Args& p, char* smem, const int reg, const int rb, const int tm, const int tn,
                                          const int tm_img) {
    char (*raw)[RAWB] = reinterpret_cast<char (*)[RAWB]>(smem);
    float* const ex = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int xi = tid >> 6;
    const int TX = p.TX[reg], TY = p.TY[reg];
    const int byi = fdiv(rb, p.bx[reg], p.m_bx[reg]);
    const int bxi = rb - byi * p.bx[reg];
    const int tys = p.tiles_y, pitch = 2 * tys + 2;
    const int R0 = tm * TY;                                // stacked mode: first tile row (global) of this block
    const int n_img = p.stack ? fdiv(R0, tys, p.m_tys) : tm_img;   // first image of this block
    const int prow0 = p.stack ? 2 * (R0 - n_img * tys) : 0;
    const int n_here = min(p.G, p.N - n_img);
    const int ty0 = p.stack ? 0 : p.ry0[reg] + byi * TY, tx0 = p.stack ? 0 : p.rx0[reg] + bxi * TX;
    const int tye = p.rye[reg], txe = p.rxe[reg];
    const int iy0 = 2 * ty0 - p.pad, ix0 = 4 * tx0 - p.pad; // top-left input pixel of the raw patch (regions mode)

    const int PW = 4 * TX + 2, PH = p.stack ? p.PH : 2 * TY + 2;
    const int NPIX = PW * PH;
    const int HS = HSC ? HSC : ((TX + 1) | 1);             // 16-byte entries per (patch row, x mod 4); odd: the four x phases of
                                                           // consecutive pixels land in four different 16-byte bank groups (stores)
    // A pair of patch rows (= one tile row step) takes PR entries, PR = 8 * HS rounded up to TX (mod 8): tile (r, tx) then
    // sits at r * PR + tx = lane (mod 8) -- the eight lanes of a ds_read_b128 phase always hit eight different 16-byte
    // bank groups, whatever the patch shape (without it tile rows alias: 8 * HS = 0 mod 8, two-way conflicts for TX < 8)
    const int PR = 8 * HS + (TX & 7);
    const int NENT0 = PR * ((PH + 1) >> 1);                // entries per channel half ...
    const int NENT = NENT0 + ((4 - NENT0) & 7);            // ... placed 4 (mod 8) apart: the two halves of a pixel (adjacent loader
                                                           // lanes) never share a bank group
    const int nslots = TX * TY;
    auto slot = [&](int id, int& g, int& ty, int& tx, int& prow) -> bool {
        const int r = fdiv(id, TX, p.m_TX[reg]);
        tx = tx0 + (id - r * TX);
        if (p.stack) {
            const int R = R0 + r;
            const int n = fdiv(R, tys, p.m_tys);
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
        const int v = fdiv(pix, PW, p.m_PW[reg]);          // patch row
        const int px = pix - v * PW;
        int g = 0, gy = iy0 + v;
        if (p.stack) {
            const int vr = prow0 + v;
            g = fdiv(vr, pitch, p.m_pitch);
            gy = vr - g * pitch - p.pad;
        }
        const int gx = ix0 + px;
        const bool inb = ok && g < n_here && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        goff[i] = inb ? (unsigned)((((g * p.H + gy) * p.W + gx) * p.C + half * 4) * 4) : kOob;
        loff[i] = ok ? (half * NENT + (v >> 1) * PR + ((v & 1) * 4 + (px & 3)) * HS + (px >> 2)) * 16 : 2 * ENTMAX * 16;
    }
    constexpr int NRS = NT == 1 ? 2 : 1;     // register sets of the raw patch: prefetch distance 2 chunks / 1 (twice as long) chunk
    f32x4 rset[NRS][NI];
    auto load_raw = [&](f32x4 (&rs)[NI], int chunk) {
        // chunks past the end (the prefetch runs 2-4 ahead) re-read the last chunk and are never used (one s_min; unclamped they
        // would read the next pixel's channels or, at the image group's last pixel, fall outside the descriptor and read zeros:
        // gfx950 range-checks the scalar offset too, profiles/r06_soffset_probe.txt)
        const int so = min(chunk, p.nchunks - 1) * 32;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            rs[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, goff[i], so, 0));
    };
    auto store_raw = [&](const f32x4 (&rs)[NI], int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(&raw[buf][loff[i]]) = rs[i];
    };

    // ---- weight fragments: [tn32][chunk][p = 6*xi + nu][lane][4]; this block's n-tiles are tn32 = NT * tn + nt --------
    const int ntile_bytes = p.nchunks * 24576;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.u + (size_t)tn * NT * ntile_bytes), 0, NT * ntile_bytes, 0x00020000);
    // lane offsets of positions nu = 0..3 and 4..5: the per-position 1 KiB steps then fit the instruction's 12-bit immediate, and the
    // scalar offset is one value per chunk (and n-tile).  Chunks past the end are clamped to the last one and never used.
    // (ONE lane offset, made opaque once per chunk: hipcc otherwise hoists the six "offset + nu KiB" sums out of the K loop, parks
    //  them in AccVGPRs and pays a v_accvgpr_read -- a vector-ALU instruction, i.e. ~20 idle cycles of the fp32 matrix pipe -- per load)
    int uoff0 = (xi * 6 * 64 + lane) * 16;
    constexpr int NBS = 2;                   // weight register sets: chunk t + 1's are requested at the top of chunk t
    f32x4 bfs[NBS][NT][6];
    auto load_b = [&](f32x4 (&bf)[NT][6], int nu, int chunk) {       // position nu of every n-tile
        const int so = min(chunk, p.nchunks - 1) * 24576 + (nu >> 2) * 4096;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            bf[nt][nu] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, uoff0 + (nu & 3) * 1024, so + nt * ntile_bytes, 0));
    };

    // ---- input transform --------------------------------------------------------------------------------------------
    // rows (B2t, this wave's xi): xi0: d0 - d2, xi1: d1 + d2, xi2: d2 - d1, xi3: d1 - d3   =>  T_j = d[ra][j] + cb * d[rb][j]
    // columns (B4t over j = 0..5):
    //   V0 = 4 T0 - 5 T2 + T4            V5 = 4 T1 - 5 T3 + T5
    //   V1 = (T4 - 4 T2) + (T3 - 4 T1)   V2 = (T4 - 4 T2) - (T3 - 4 T1)
    //   V3 = (T4 - T2) + 2 (T3 - T1)     V4 = (T4 - T2) - 2 (T3 - T1)
    const int ra = xi == 0 ? 0 : xi == 2 ? 2 : 1;
    const int rbw = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
    const float cb = __builtin_amdgcn_readfirstlane(xi) == 1 ? 1.f : -1.f;      // wave-uniform: lives in an SGPR
    int rbase;
    {
        int g, ty, tx, prow;
        if (!slot(lane & 31, g, ty, tx, prow)) slot(0, g, ty, tx, prow);    // idle slots read tile 0 (never stored)
        rbase = ((lane >> 5) * NENT + (prow >> 1) * PR + (tx - tx0)) * 16;      // prow is even
    }
    const int oa = ((ra >> 1) * PR + (ra & 1) * 4 * HS) * 16, ob = ((rbw >> 1) * PR + (rbw & 1) * 4 * HS) * 16;
    const int c1 = HS * 16;                                 // column j: phase (j & 3) at entry (j >> 2)
    const f32x2 cb2 = {cb, cb};
    const f32x2 k4 = {4.f, 4.f}, km5 = {-5.f, -5.f}, km4 = {-4.f, -4.f}, k2 = {2.f, 2.f}, km2 = {-2.f, -2.f};
    f32x4 va[6];            // A fragments of the current chunk (rolling: overwritten pair by pair with the next chunk's)
    f32x4 vb[2];            // NT = 2: the second set of positions 3, 4 (chunks alternate between va[3..4] and vb[0..1])
    f32x4 T0, T1, T2, T3, T4, T5;
    f32x4 xa[3], xb[3];
    f32x4 ya[3], yb[3];     // NT = 2: the odd columns, so that all twelve reads of a chunk are in flight together
    auto rdA = [&](int buf) {            // columns 0, 2, 4
        const char* base = &raw[buf][rbase];
        xa[0] = *reinterpret_cast<const f32x4*>(base + oa);
        xb[0] = *reinterpret_cast<const f32x4*>(base + ob);
        xa[1] = *reinterpret_cast<const f32x4*>(base + oa + 2 * c1);
        xb[1] = *reinterpret_cast<const f32x4*>(base + ob + 2 * c1);
        xa[2] = *reinterpret_cast<const f32x4*>(base + oa + 16);
        xb[2] = *reinterpret_cast<const f32x4*>(base + ob + 16);
    };
    auto rdB = [&](int buf) {            // columns 1, 3, 5
        const char* base = &raw[buf][rbase];
        xa[0] = *reinterpret_cast<const f32x4*>(base + oa + c1);
        xb[0] = *reinterpret_cast<const f32x4*>(base + ob + c1);
        xa[1] = *reinterpret_cast<const f32x4*>(base + oa + 3 * c1);
        xb[1] = *reinterpret_cast<const f32x4*>(base + ob + 3 * c1);
        xa[2] = *reinterpret_cast<const f32x4*>(base + oa + c1 + 16);
        xb[2] = *reinterpret_cast<const f32x4*>(base + ob + c1 + 16);
    };
    auto rdB2 = [&](int buf) {           // columns 1, 3, 5 into their own registers
        const char* base = &raw[buf][rbase];
        ya[0] = *reinterpret_cast<const f32x4*>(base + oa + c1);
        yb[0] = *reinterpret_cast<const f32x4*>(base + ob + c1);
        ya[1] = *reinterpret_cast<const f32x4*>(base + oa + 3 * c1);
        yb[1] = *reinterpret_cast<const f32x4*>(base + ob + 3 * c1);
        ya[2] = *reinterpret_cast<const f32x4*>(base + oa + c1 + 16);
        yb[2] = *reinterpret_cast<const f32x4*>(base + ob + c1 + 16);
    };
    auto cTB2 = [&]() { T1 = fma4s(cb2, yb[0], ya[0]); T3 = fma4s(cb2, yb[1], ya[1]); T5 = fma4s(cb2, yb[2], ya[2]); };
    auto cV34to = [&](f32x4& v3, f32x4& v4) {
        const f32x4 c = sub4(T4, T2), d = sub4(T3, T1);
        v3 = fma4s(k2, d, c);
        v4 = fma4s(km2, d, c);
    };
    auto cTA = [&]() { T0 = fma4s(cb2, xb[0], xa[0]); T2 = fma4s(cb2, xb[1], xa[1]); T4 = fma4s(cb2, xb[2], xa[2]); };
    auto cTB = [&]() { T1 = fma4s(cb2, xb[0], xa[0]); T3 = fma4s(cb2, xb[1], xa[1]); T5 = fma4s(cb2, xb[2], xa[2]); };
    auto cV05 = [&]() {
        va[0] = fma4s(km5, T2, fma4s(k4, T0, T4));
        va[5] = fma4s(km5, T3, fma4s(k4, T1, T5));
    };
    auto cV12 = [&]() {
        const f32x4 a = fma4s(km4, T2, T4), bq = fma4s(km4, T1, T3);
        va[1] = add4(a, bq);
        va[2] = sub4(a, bq);
    };
    auto cV34 = [&]() {
        const f32x4 c = sub4(T4, T2), d = sub4(T3, T1);
        va[3] = fma4s(k2, d, c);
        va[4] = fma4s(km2, d, c);
    };

    f32x16 acc[6][NT];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][nt][r] = 0.f;

#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(nu, kk) do { _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
        acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[nu][kk], bcur[nt][nu][kk], acc[nu][nt], 0, 0, 0); if (nt + 1 < NT) SB(); } } while (0)
#define MG(frag, nu, kk) do { _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
        acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[kk], bcur[nt][nu][kk], acc[nu][nt], 0, 0, 0); if (nt + 1 < NT) SB(); } } while (0)

    // ---- prologue -------------------------------------------------------------------------------------------------
    load_raw(rset[0], 0);
    if constexpr (NRS == 2) load_raw(rset[1], 1);
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) load_b(bfs[0], nu, 0);
    store_raw(rset[0], 0);
    if constexpr (NRS == 2) {
        store_raw(rset[1], 1);
        load_raw(rset[0], 2);
        load_raw(rset[1], 3);
    } else {
        load_raw(rset[0], 1);
        store_raw(rset[0], 1);
        load_raw(rset[0], 2);
    }
    __syncthreads();
    rdA(0); cTA(); rdB(0); cTB(); cV05(); cV12();     // NT = 1: V3, V4 of chunk 0 are made in its first slot
    if constexpr (NT == 2) cV34to(va[3], va[4]);
    __syncthreads();                                  // chunk 0 overwrites raw[0] right away

    // At the top of chunk t: raw[t&1] = patch(t) (already consumed), raw[(t+1)&1] = patch(t+1), rset[t&1] = patch(t+2) in
    // flight, rset[(t+1)&1] = patch(t+3) in flight, bf = weights(t), va[0,5,1,2] = A fragments of chunk t, T0..T5 = the
    // row transform of chunk t (V3, V4 still to be made from it).
#define A1(x) do { if (!(SEAM_W24_ABL & 1)) { x; } } while (0)
#define A2(x) do { if (!(SEAM_W24_ABL & 2)) { x; } } while (0)
#define A8(x) do { if (!(SEAM_W24_ABL & 8)) { x; } } while (0)
    // Weights: two register sets, all six loads of chunk t+1 issued at the top of chunk t, BEFORE the patch loads -- the
    // vector-memory counter retires in order, so a wait for a weight fragment also waits for every older load: with the
    // weights first, the (HBM-latency) patch loads of chunk t are not forced to complete before the top of chunk t+2.
    auto chunk = [&](int t, int par, f32x4 (&bcur)[NT][6], f32x4 (&bnext)[NT][6]) {
        asm volatile("" : "+v"(uoff0));
        if constexpr (NT == 1) {
            SB(); MF(0, 0); A8(cV34());
            SB(); MF(5, 0); A8(rdA(par ^ 1));
            SB(); MF(0, 1);
            SB(); MF(5, 1);
            SB(); MF(0, 2); A8(cTA());
            SB(); MF(5, 2); A8(rdB(par ^ 1));
            SB(); MF(0, 3);
            SB(); MF(5, 3);
            SB(); MF(1, 0); A8(cTB());
            SB(); MF(2, 0); A8(cV05());
            SB(); MF(1, 1); A2(load_b(bnext, 0, t + 1); load_b(bnext, 5, t + 1));
            SB(); MF(2, 1); A2(load_b(bnext, 1, t + 1); load_b(bnext, 2, t + 1));
            SB(); MF(1, 2); A2(load_b(bnext, 3, t + 1); load_b(bnext, 4, t + 1));
            SB(); MF(2, 2);
            SB(); MF(1, 3);
            SB(); MF(2, 3);
            SB(); MF(3, 0); A8(cV12());
            SB(); MF(4, 0); A1(store_raw(rset[par], par));
            SB(); MF(3, 1); A1(load_raw(rset[par], t + 4));
            SB(); MF(4, 1);
            SB(); MF(3, 2);
            SB(); MF(4, 2);
            SB(); MF(3, 3);
            SB(); MF(4, 3);
            SB();
        } else {
            // Two MFMAs per MF().  What tools/mfma_shadow_probe.hip measured (profiles/r04_mfma_shadow_probe.txt): in ONE wave no
            // vector-ALU instruction overlaps an fp32 MFMA -- a group of k of them between two MFMAs idles the matrix pipe for
            // ~17 + 4.5 k cycles -- while LDS reads, global loads and scalar instructions are free.  So the whole input
            // transform of chunk t + 1 (36 packed ops) is ONE group, placed behind the 32 MFMAs of positions 0, 5, 1, 2 (whose
            // fragments it overwrites); positions 3, 4 alternate between two register sets (va[3..4] / vb[0..1] by chunk
            // parity), so nothing of the transform is carried across the chunk boundary; the twelve patch reads it needs are
            // issued together, 8 MFMAs ahead.  Two weight register sets (the arch file has room: 162 of 256 were in use): chunk
            // t + 1's twelve fragments are requested at the top of chunk t, ahead of the patch loads in the in-order vmcnt queue.
            SB(); MF(0, 0); A2(load_b(bnext, 0, t + 1));
            SB(); MF(5, 0); A2(load_b(bnext, 5, t + 1));
            SB(); MF(0, 1); A2(load_b(bnext, 1, t + 1));
            SB(); MF(5, 1); A2(load_b(bnext, 2, t + 1));
            SB(); MF(0, 2); A2(load_b(bnext, 3, t + 1));
            SB(); MF(5, 2); A2(load_b(bnext, 4, t + 1));
            SB(); MF(0, 3);
            SB(); MF(5, 3);
            SB(); MF(1, 0);
            SB(); MF(2, 0);
            SB(); MF(1, 1); A1(store_raw(rset[0], par));
            SB(); MF(2, 1); A1(load_raw(rset[0], t + 3));
            SB(); MF(1, 2); A8(rdA(par ^ 1); rdB2(par ^ 1));
            SB(); MF(2, 2);
            SB(); MF(1, 3);
            SB(); MF(2, 3);
            SB();
            A8(cTA(); cTB2(); cV05(); cV12());
            if (par == 0) {
                A8(cV34to(vb[0], vb[1]));
                SB(); MG(va[3], 3, 0);
                SB(); MG(va[4], 4, 0);
                SB(); MG(va[3], 3, 1);
                SB(); MG(va[4], 4, 1);
                SB(); MG(va[3], 3, 2);
                SB(); MG(va[4], 4, 2);
                SB(); MG(va[3], 3, 3);
                SB(); MG(va[4], 4, 3);
            } else {
                A8(cV34to(va[3], va[4]));
                SB(); MG(vb[0], 3, 0);
                SB(); MG(vb[1], 4, 0);
                SB(); MG(vb[0], 3, 1);
                SB(); MG(vb[1], 4, 1);
                SB(); MG(vb[0], 3, 2);
                SB(); MG(vb[1], 4, 2);
                SB(); MG(vb[0], 3, 3);
                SB(); MG(vb[1], 4, 3);
            }
            SB();
        }
        if (!(SEAM_W24_ABL & 4)) __syncthreads();
    };
    for (int t = 0; t < p.nchunks; t += 2) {
        chunk(t, 0, bfs[0], bfs[NBS - 1]);
        if (t + 1 < p.nchunks) chunk(t + 1, 1, bfs[NBS - 1], bfs[0]);
    }
#undef SB
#undef MF
#undef MG
#undef A1
#undef A2
#undef A8

    // ---- epilogue: output transform + scale/shift (+ residual, ReLU) ------------------------------------------------
    //   columns (A4t over nu): Y0 = m0+m1+m2+m3+m4, Y1 = (m1-m2) + 2(m3-m4), Y2 = (m1+m2) + 4(m3+m4), Y3 = (m1-m2) + 8(m3-m4) + m5
    //   rows (A2t over xi, across the waves): y0 = q0+q1+q2, y1 = q1-q2-q3
    if (SEAM_W24_ABL & 16) {            // experiment: no epilogue (one conditional store keeps the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int nu = 0; nu < 6; ++nu)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[nu][nt][r];
        if (sum == 123456.789f) p.y[tid] = sum;
        return;
    }
    const size_t out_img = (size_t)p.Ho * p.Wo * p.K * 4;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)p.y + (size_t)n_img * out_img), 0, (int)(out_img * n_here), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(p.res ? p.res : p.y) + (size_t)n_img * out_img), 0, (int)(out_img * n_here), 0x00020000);
    const int et = tid >> 3;          // tile of the exchange this thread finishes
    const int n4 = tid & 7;
    int g, tyt, txt, prow_unused;
    const bool tile_ok = slot(et, g, tyt, txt, prow_unused) && g < n_here;
    const int oy = 2 * tyt, ox = 4 * txt;
    const int cstep = p.K * 4, rstep = p.Wo * cstep;                      // bytes per output pixel / row
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ncol = (tn * NT + nt) * 32 + n4 * 4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + ncol);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + ncol);
        const unsigned obase = (unsigned)(((g * p.Ho + oy) * p.Wo + ox) * p.K + ncol) * 4u;
        __syncthreads();              // previous readers of `ex` (first pass: of the raw buffers it aliases) are done
        // nu half (A4t) on register PAIRS (acc[nu][nt][r], [r+1] are adjacent registers): packed fp32; each pair goes straight to LDS
        {
            const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c8 = {8.f, 8.f};
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                // NT = 2: the 192 accumulators live in AccVGPRs; read the 12 values of this step explicitly, one step at a time --
                // left to itself the compiler copies ALL of them into arch VGPRs at the loop exit and pays for that register peak
                // by spilling the K loop's address registers
                auto rd = [&](int nu) -> f32x2 {
                    if constexpr (NT == 2) {
                        float x0, x1;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(acc[nu][nt][2 * h]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(acc[nu][nt][2 * h + 1]));
                        return f32x2{x0, x1};
                    } else {
                        return f32x2{acc[nu][nt][2 * h], acc[nu][nt][2 * h + 1]};
                    }
                };
                const f32x2 m0 = rd(0), m1 = rd(1), m2 = rd(2), m3 = rd(3), m4 = rd(4), m5 = rd(5);
                const f32x2 s12 = pk_add(m1, m2), d12 = pk_sub(m1, m2), s34 = pk_add(m3, m4), d34 = pk_sub(m3, m4);
                const f32x2 y0 = pk_add(pk_add(m0, s12), s34);
                const f32x2 y1 = pk_fma_s(c2, d34, d12);
                const f32x2 y2 = pk_fma_s(c4, s34, s12);
                const f32x2 y3 = pk_add(pk_fma_s(c8, d34, d12), m5);
                const int r = 2 * h;
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float* e0 = &ex[((xi * 4 + 0) * 32 + row) * 32 + (lane & 31)];
                e0[0] = y0[0];               e0[32] = y0[1];
                e0[1024] = y1[0];            e0[1024 + 32] = y1[1];
                e0[2048] = y2[0];            e0[2048 + 32] = y2[1];
                e0[3072] = y3[0];            e0[3072 + 32] = y3[1];
            }
        }
        __syncthreads();
#pragma unroll
        for (int bcol = 0; bcol < 4; ++bcol) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(&ex[((0 * 4 + bcol) * 32 + et) * 32 + n4 * 4]);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(&ex[((1 * 4 + bcol) * 32 + et) * 32 + n4 * 4]);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(&ex[((2 * 4 + bcol) * 32 + et) * 32 + n4 * 4]);
            const f32x4 q3 = *reinterpret_cast<const f32x4*>(&ex[((3 * 4 + bcol) * 32 + et) * 32 + n4 * 4]);
            f32x4 yv[2];
            yv[0] = q0 + q1 + q2;
            yv[1] = q1 - q2 - q3;
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                const bool ok = tile_ok && (oy + aa) < p.Ho && (ox + bcol) < p.Wo;
                const unsigned off = ok ? obase + (unsigned)(aa * rstep + bcol * cstep) : kOob;
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

template <int NT>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void conv3x3_wino24(const Wino24Args p) {
    // two raw patch buffers; the epilogue's exchange array ex[xi][b][tile][n] (64 KiB, all four output columns of a tile in one
    // pass) reuses the same memory after the K loop
    constexpr int EXB = 4 * 4 * 32 * 32 * 4;
    static_assert(2 * RAWB <= EXB, "the raw buffers live inside the exchange array's 64 KiB");
    __shared__ __attribute__((aligned(16))) char smem[EXB];

    // ---- XCD-aware tile id (bijective) ----------------------------------------------------------------------------
    const int nblk = gridDim.x;
    const int b = blockIdx.x;
    const int xcd = b & 7;
    int tm, tn;
    if (p.nsplit > 1) {
        // The n-tiles are split over `nsplit` groups of XCDs (workgroup b runs on XCD b & 7): XCD x streams only the n-tile subset
        // g = x % nsplit -- its share of the transformed weights (24*K*C*4 / nsplit bytes) stays in the XCD's 4 MiB L2 instead of
        // being re-streamed from the Infinity Cache by every generation of resident blocks, so the latency-critical weight loads
        // hit -- and the 8 / nsplit XCDs of a group share the patch range of partition pi = x / nsplit (every patch is then read
        // by nsplit XCDs: patch loads run 2+ chunks ahead and tolerate the miss).
        const int g = xcd % p.nsplit, pi = xcd / p.nsplit;
        const int j = b >> 3;
        const int tml = fdiv(j, p.tns, p.m_tns);
        tn = g * p.tns + (j - tml * p.tns);
        const int size = p.part_q + (pi < p.part_r ? 1 : 0);
        if (tml >= size) return;                           // padding blocks of the shorter partitions
        tm = pi * p.part_q + min(pi, p.part_r) + tml;
    } else {
        const int q8 = nblk >> 3, rem8 = nblk & 7;
        const int tile = (xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8) + (b >> 3);
        tm = fdiv(tile, p.tiles_n, p.m_tiles_n);
        tn = tile - tm * p.tiles_n;
    }
    const int per_img = p.per_img;
    const int tm_img = fdiv(tm, per_img, p.m_per_img);
    int rb = tm - tm_img * per_img;
    int reg = 0;
    if (p.nreg > 1 && rb >= p.bx[0] * p.by[0]) {
        rb -= p.bx[0] * p.by[0];
        reg = 1;
        if (p.nreg > 2 && rb >= p.bx[1] * p.by[1]) { rb -= p.bx[1] * p.by[1]; reg = 2; }
    }
    const int hs = (p.TX[reg] + 1) | 1;                    // block-uniform
    if (hs == 9) w24_block<NT, 9>(p, smem, reg, rb, tm, tn, tm_img);
    else if (hs == 3) w24_block<NT, 3>(p, smem, reg, rb, tm, tn, tm_img);
    else if (hs == 5) w24_block<NT, 5>(p, smem, reg, rb, tm, tn, tm_img);
    else w24_block<NT, 0>(p, smem, reg, rb, tm, tn, tm_img);
}

// =====================================================================================================================
// conv3x3_wino24pc (round 5): the NT = 2 block as a PRODUCER / CONSUMER pair of wave groups, two waves per SIMD, persistent.
//
// What round 4 measured on conv3x3_wino24<2>: the block is alone on its CU (one wave per SIMD, 490 registers), so everything that
// is not an MFMA is exposed -- the input transform, the patch loads / LDS stores, the waits behind them: ~17 % of every chunk plus a
// fixed 5.3 us per block.  What tools/probes/pc_probe.hip measured this round: beside a wave that streams fp32 MFMAs back to back,
// its SIMD partner's LDS, vector-memory and scalar instructions are free, but its VECTOR-ALU instructions do not issue at all until
// the matrix pipe idles (fp32 MFMA and fp32 VALU share the SIMD's FMA lanes).  So the work is split by role, and the one thing
// that has to share the pipe -- the 36 packed ops of a chunk's input transform -- is placed where the consumer yields anyway:
//   waves 0..3 (consumers, one per SIMD): MFMAs only.  Wave xi owns the positions (xi, nu = 0..5) x 32 tiles x 64 channels
//     (192 accumulators, all arch VGPRs: with 512 threads per block the budget is 256 registers and hipcc selects the VGPR
//     form of the MFMA).  A fragments come from LDS (`ds_read_b128`, one position ahead), B fragments straight from global
//     memory through a register ring (RING positions x 8 registers, refilled right behind the MFMAs that read them); the K loop
//     holds no vector-ALU instruction; the first k-step of a tile multiplies into a zero constant (no accumulator clears).
//   waves 4..7 (producers, the SIMD partners): a software pipeline over the block's chunk stream, one interval per consumer
//     chunk -- fragments of chunk t + 2 registers -> LDS V[t & 1]; raw patch of chunk t + 4 registers -> LDS raw[t & 1]; global
//     loads of chunk t + 6; LDS reads of chunk t + 3's raw patch (all free beside the MFMAs); then the 36 packed ops of chunk
//     t + 3, which run while the consumer waits at the chunk's barrier; then the barrier.  The pipeline runs ACROSS tiles: a block
//     is persistent (grid = CUs, XCD-contiguous tile ranges), the next tile's address set-up is computed nine intervals before
//     the current tile ends and the load / store / read stages switch over one by one, so a tile's prologue (first patch
//     latency, two transforms) disappears behind the previous tile's last chunks and epilogue.
//   One s_barrier per chunk, in the consumer between positions 4 and 5: it then holds the chunk's last fragment in registers
//     and requests the next chunk's first one under the eight MFMAs of position 5.
//   Epilogue: consumers run the nu half of the output transform into the exchange array (its own 64 KiB: the raw / V buffers hold
//     the next tile's data by then), all 512 threads finish (xi half, scale / shift / residual / ReLU, 16-byte stores).
// The arithmetic (transform expressions, accumulation order, epilogue) is that of conv3x3_wino24<2>: results are bit-identical.
// =====================================================================================================================
#ifndef SEAM_W24PC_ABL
#define SEAM_W24PC_ABL 0    // experiments: 1 no in-loop patch loads / stores (64: no stores only, 128: no requests only), 2 no in-loop weight loads, 8 no in-loop transforms, 16 no epilogue arithmetic,
                            // 256 (round 6, TIMING ONLY -- results are garbage): the weight fragments travel through LDS -- the producers issue
                            // them as LDS-DMA loads (buffer_load_dwordx4 ... lds) into the exchange region, the consumers read them with ds_read_b128
#endif
#ifndef SEAM_W24PC_RING
#define SEAM_W24PC_RING 4
#endif
#ifndef SEAM_W24PC_AD
#define SEAM_W24PC_AD 1
#endif
constexpr int PC_VB = 4 * 6 * 64 * 16;                 // bytes per V buffer
constexpr int PC_RAW = 0, PC_V = 2 * RAWB, PC_EX = 2 * RAWB + 2 * PC_VB;      // LDS map: raw[2] | V[2] | ex
constexpr int PC_EXB = 4 * 4 * 32 * 32 * 4;
constexpr int PC_LDS = PC_EX + PC_EXB;
static_assert((2 * RAWB) % 16 == 0 && PC_LDS <= 160 * 1024, "LDS map");

#define PC_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define LDSQ __attribute__((address_space(3)))
#ifdef SEAM_W24PC_TRACE      // debug: stamp (tag << 56 | s_memtime) into p.trace[wave * 4096 + k++] from lane 0 of the traced block
#define PC_TR(tag) do { if (tr_on) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if (lane == 0 && tr_k < 4096) p.trace[wave * 4096 + tr_k] = tm_ | ((unsigned long long)(tag) << 56); ++tr_k; } } while (0)
#else
#define PC_TR(tag) do { } while (0)
#endif

// wave-uniform geometry of one tile (SGPRs; recomputed from the tile index where it is needed rather than kept alive)
struct PcGeo {
    int tn, reg, n_img, n_here, R0, prow0, ty0, tx0, iy0, ix0, TX, PW, NPIX, HS, PR, NENT, nslots, tye, txe;
    unsigned m_TX, m_PW;
};
__device__ __forceinline__ PcGeo pc_geo(const Wino24Args& p, const int tile) {
    PcGeo g;
    const int tm = fdiv(tile, p.tiles_n, p.m_tiles_n);
    g.tn = tile - tm * p.tiles_n;
    const int tm_img = fdiv(tm, p.per_img, p.m_per_img);
    int rb = tm - tm_img * p.per_img;
    int reg = 0;
    const int c0 = p.rg[0].bx * p.rg[0].by, c1 = p.rg[1].bx * p.rg[1].by;
    if (p.nreg > 1 && rb >= c0) {
        rb -= c0;
        reg = 1;
        if (p.nreg > 2 && rb >= c1) { rb -= c1; reg = 2; }
    }
    g.reg = reg;
    const Wino24Args::Rg R = p.rg[reg];         // one 64-byte scalar load
    const int TX = R.TX, TY = R.TY;
    const int byi = fdiv(rb, R.bx, R.m_bx);
    const int bxi = rb - byi * R.bx;
    g.R0 = tm * TY;
    g.n_img = p.stack ? fdiv(g.R0, p.tiles_y, p.m_tys) : tm_img;
    g.prow0 = p.stack ? 2 * (g.R0 - g.n_img * p.tiles_y) : 0;
    g.n_here = min(p.G, p.N - g.n_img);
    g.ty0 = p.stack ? 0 : R.ry0 + byi * TY;
    g.tx0 = p.stack ? 0 : R.rx0 + bxi * TX;
    g.iy0 = 2 * g.ty0 - p.pad;
    g.ix0 = 4 * g.tx0 - p.pad;
    g.TX = TX;
    g.PW = 4 * TX + 2;
    const int PH = p.stack ? p.PH : 2 * TY + 2;
    g.NPIX = g.PW * PH;
    g.HS = (TX + 1) | 1;
    g.PR = 8 * g.HS + (TX & 7);
    const int NENT0 = g.PR * ((PH + 1) >> 1);
    g.NENT = NENT0 + ((4 - NENT0) & 7);
    g.nslots = TX * TY;
    g.m_TX = R.m_TX; g.m_PW = R.m_PW; g.tye = R.rye; g.txe = R.rxe;
    return g;
}
// tile slot id (0..31) of a block patch -> image-in-group g, tile row / column, first patch row; false for idle slots
__device__ __forceinline__ bool pc_slot(const Wino24Args& p, const PcGeo& q, int id, int& g, int& ty, int& tx, int& prow) {
    const int r = fdiv(id, q.TX, q.m_TX);
    tx = q.tx0 + (id - __mul24(r, q.TX));
    if (p.stack) {
        const int R = q.R0 + r;
        const int n = fdiv(R, p.tiles_y, p.m_tys);
        g = n - q.n_img;
        ty = R - __mul24(n, p.tiles_y);
        prow = __mul24(2 * p.tiles_y + 2, g) + 2 * ty - q.prow0;
        return id < q.nslots && n < p.N;
    }
    g = 0;
    ty = q.ty0 + r;
    prow = 2 * r;
    return id < q.nslots && ty < q.tye && tx < q.txe;
}

// second half of the epilogue for one n-tile (all 512 threads; consumers take the output columns 0, 1 of a tile, producers 2, 3).
// sc / sh: this thread's four channels of the epilogue vectors, requested by the caller ahead of the barrier in front of this call.
struct PcEpi { f32x4 sc[2], sh[2]; };
__device__ __forceinline__ void pc_epi_vectors(const Wino24Args& p, const PcGeo& q, const int tid, PcEpi& e) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int ncol = (q.tn * 2 + nt) * 32 + (tid & 7) * 4;
        e.sc[nt] = f32x4{1.f, 1.f, 1.f, 1.f};
        e.sh[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.scale) e.sc[nt] = *reinterpret_cast<const f32x4*>(p.scale + ncol);
        if (p.shift) e.sh[nt] = *reinterpret_cast<const f32x4*>(p.shift + ncol);
    }
}
__device__ __forceinline__ void pc_finish(const Wino24Args& p, const PcGeo& q, const float* ex, const int nt, const int tid, const bool consumer,
                                          const PcEpi& epi) {
    const int et = (tid & 255) >> 3;
    const int n4 = tid & 7;
    int g, tyt, txt, prow_unused;
    const bool tile_ok = pc_slot(p, q, et, g, tyt, txt, prow_unused) && g < q.n_here;
    const int oy = 2 * tyt, ox = 4 * txt;
    const int cstep = p.K * 4, rstep = p.Wo * cstep;
    const size_t out_img = (size_t)p.Ho * p.Wo * p.K * 4;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)p.y + (size_t)q.n_img * out_img), 0, (int)(out_img * q.n_here), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(p.res ? p.res : p.y) + (size_t)q.n_img * out_img), 0, (int)(out_img * q.n_here), 0x00020000);
    const int ncol = (q.tn * 2 + nt) * 32 + n4 * 4;
    const f32x4 sc = epi.sc[nt], sh = epi.sh[nt];
    // (operands below 2^24: the launcher caps an image group's output at 2^31 bytes and K >= 64 -- the full-rate 24-bit multiply)
    const unsigned obase = (unsigned)(__mul24(__mul24(__mul24(g, p.Ho) + oy, p.Wo) + ox, p.K) + ncol) * 4u;
    unsigned off[2][2];
    f32x4 rv[2][2];
#pragma unroll
    for (int bc = 0; bc < 2; ++bc)
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) {
            const int bcol = (consumer ? 0 : 2) + bc;
            const bool ok = tile_ok && (oy + aa) < p.Ho && (ox + bcol) < p.Wo;
            off[bc][aa] = ok ? obase + (unsigned)(aa * rstep + bcol * cstep) : kOob;
            if (p.res) rv[bc][aa] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off[bc][aa], 0, 0));   // all four ahead of the LDS reads
        }
#pragma unroll
    for (int bc = 0; bc < 2; ++bc) {
        const int bcol = (consumer ? 0 : 2) + bc;
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(&ex[((0 * 4 + bcol) * 32 + et) * 32 + (n4 ^ (et & 7)) * 4]);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(&ex[((1 * 4 + bcol) * 32 + et) * 32 + (n4 ^ (et & 7)) * 4]);
        const f32x4 q2 = *reinterpret_cast<const f32x4*>(&ex[((2 * 4 + bcol) * 32 + et) * 32 + (n4 ^ (et & 7)) * 4]);
        const f32x4 q3 = *reinterpret_cast<const f32x4*>(&ex[((3 * 4 + bcol) * 32 + et) * 32 + (n4 ^ (et & 7)) * 4]);
        f32x4 yv[2];
        yv[0] = q0 + q1 + q2;
        yv[1] = q1 - q2 - q3;
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) {
            f32x4 v = yv[aa] * sc + sh;
            if (p.res) {
                if (p.relu == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = rv[bc][aa][e] > 0.f ? v[e] : 0.f;
                } else {
                    v += rv[bc][aa];
                }
            }
            if (p.relu == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc, off[bc][aa], 0, 0);
        }
    }
}

template <int RING>
__global__ __launch_bounds__(512, 2) void conv3x3_wino24pc(const Wino24Args p) {
    constexpr int NT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char (*raw)[RAWB] = reinterpret_cast<char (*)[RAWB]>(smem + PC_RAW);
    float* const ex = reinterpret_cast<float*>(smem + PC_EX);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int xi = wave & 3;
    const int n = p.nchunks;

    // ---- the block's tiles: XCD x (= blockIdx & 7) owns a contiguous range of the launch's tiles; its blocks walk it interleaved ----
    const int T = p.total_tiles;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, sl0 = blockIdx.x >> 3;
    const int q8 = T >> 3, rem8 = T & 7;
    const int cnt = q8 + (xcd < rem8 ? 1 : 0);
    const int start = xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8;
    const int S = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
    const int ntiles = sl0 < cnt ? (cnt - sl0 + S - 1) / S : 0;
    if (ntiles == 0) return;
    const int tile0 = start + sl0;

    const int vlane = PC_V + (xi * 6 * 64 + lane) * 16;        // this wave pair's row of V[0]: + buf * PC_VB + nu * 1024
#ifdef SEAM_W24PC_TRACE
    const bool tr_on = p.trace && blockIdx.x == SEAM_W24PC_TRACE && (wave & 3) == 0;
    int tr_k = 0;
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)

#ifdef SEAM_W24PC_TRACE
    const unsigned long long blk_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (!consumer) {
        // =================================================== producer ===================================================
        const int ptid = tid - 256;
        const size_t img_bytes = (size_t)p.H * p.W * p.C * 4;
        constexpr int NP = 12;                  // 16-byte patch pieces per thread and GROUP of four chunks (one 128-byte line per pixel)
        // Address state of the three stages that touch a tile's geometry.  Each piece is recomputed for the next tile (vector ALU,
        // in a burst window) right after its stage has used it for the last time, so one set serves the whole pipeline.
        //   A patch pixel's 32 channels of a group are ONE 128-byte line: eight adjacent lanes fetch it with one 16-byte piece each
        //   (round 5: the per-chunk form -- two lanes per line, four visits per line -- missed the 32 KiB L1 every time: 4 x the
        //   L2 requests, and the vector-memory queue it kept busy held back the consumers' weight loads).
        unsigned goff[NP];                      // load stage: global byte offset of piece (ptid & 7) of pixel (ptid >> 3) + 32 r; kOob outside
        LDSQ char* lp[NP];                      // store stage: LDS address of that piece inside raw[0] (its chunk = piece >> 1, half = piece & 1)
        LDSQ char* ra[4];                       // read stage: LDS addresses (raw[0]) of the transform's two patch rows at x phases 0..3
        LDSQ char* rb[4];
        LDSQ char* lpn[NP];                     // the next tile's lp, computed together with its goff (same pixel arithmetic)
        // (this runs in a burst window, i.e. with the consumers idle: ~440 vector instructions = 3100 cycles in its general form --
        //  2.5 % of a C = 256 tile, measured with stamps.  Products have operands below 2^24 -- launcher-checked: 2^31 bytes per
        //  image group, C >= 64 -- and use the full-rate 24-bit multiply.  Consecutive tiles of a block mostly lie in the same
        //  region of a non-stacked map: their patch has the same shape, so each piece's patch coordinates (kept packed in `vp`)
        //  and LDS address are unchanged and only the image offset and the border tests are redone -- ~150 instructions.)
        int vp[NP];                             // piece r's patch row | column << 16 in the region `vp_reg` (row 0x7fff: no pixel)
        int vp_reg = -1;                        // -1: nothing cached
        auto setup_patch = [&](const PcGeo& q, LDSQ char* (&lp_)[NP]) {
            const int pitch = 2 * p.tiles_y + 2;
            if (q.reg == vp_reg) {
                if (p.stack) {
                    // stacked maps (round 6): the patch shape is the same for every tile, only the row at which the images wrap moves
                    // (prow0) -- the piece's patch coordinates and LDS address stay, the image index and its row are redone
#pragma unroll
                    for (int r = 0; r < NP; ++r) {
                        const int v = vp[r] & 0xffff, px = vp[r] >> 16;
                        const int vr = q.prow0 + v;                 // v = 0x7fff (no pixel): an image index far outside the group
                        const int g = fdiv(vr, pitch, p.m_pitch);
                        const int gy = vr - __mul24(g, pitch) - p.pad, gx = q.ix0 + px;
                        const bool inb = g < q.n_here && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                        goff[r] = inb ? (unsigned)((__mul24(__mul24(__mul24(g, p.H) + gy, p.W) + gx, p.C) + (ptid & 7) * 4) * 4) : kOob;
                    }
                    return;
                }
#pragma unroll
                for (int r = 0; r < NP; ++r) {
                    const int v = vp[r] & 0xffff, px = vp[r] >> 16;
                    const int gy = q.iy0 + v, gx = q.ix0 + px;
                    const bool inb = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;      // v = 0x7fff fails the row test
                    goff[r] = inb ? (unsigned)((__mul24(__mul24(gy, p.W) + gx, p.C) + (ptid & 7) * 4) * 4) : kOob;
                }
                return;                         // lp_ (= lpn): this region's addresses already
            }
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int pix = (ptid >> 3) + 32 * r;
                const int v = fdiv(pix, q.PW, q.m_PW);
                const int px = pix - __mul24(v, q.PW);
                int g = 0, gy = q.iy0 + v;
                if (p.stack) {
                    const int vr = q.prow0 + v;
                    g = fdiv(vr, pitch, p.m_pitch);
                    gy = vr - __mul24(g, pitch) - p.pad;
                }
                const int gx = q.ix0 + px;
                const bool inb = pix < q.NPIX && g < q.n_here && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                goff[r] = inb ? (unsigned)((__mul24(__mul24(__mul24(g, p.H) + gy, p.W) + gx, p.C) + (ptid & 7) * 4) * 4) : kOob;
                const int half = ptid & 1;
                lp_[r] = (LDSQ char*)smem + PC_RAW +
                         (pix < q.NPIX ? (__mul24(half, q.NENT) + __mul24(v >> 1, q.PR) + __mul24((v & 1) * 4 + (px & 3), q.HS) + (px >> 2)) * 16
                                       : 2 * ENTMAX * 16);
                vp[r] = pix < q.NPIX ? (v | (px << 16)) : 0x7fff;
            }
            vp_reg = q.reg;
        };
        auto setup_read = [&](const PcGeo& q) {
            // input transform of row xi: T_j = d[ra][j] + cb * d[rb][j]  (rows: xi0: d0 - d2, xi1: d1 + d2, xi2: d2 - d1, xi3: d1 - d3)
            const int rwa = xi == 0 ? 0 : xi == 2 ? 2 : 1;
            const int rwb = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
            int g, ty, tx, prow;
            if (!pc_slot(p, q, lane & 31, g, ty, tx, prow)) pc_slot(p, q, 0, g, ty, tx, prow);     // idle slots read tile 0 (never stored)
            const int rbase = (__mul24(lane >> 5, q.NENT) + __mul24(prow >> 1, q.PR) + (tx - q.tx0)) * 16;
            const int oa = ((rwa >> 1) * q.PR + (rwa & 1) * 4 * q.HS) * 16, ob = ((rwb >> 1) * q.PR + (rwb & 1) * 4 * q.HS) * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {       // column j: x phase (j & 3) at entry (j >> 2)
                ra[j] = (LDSQ char*)smem + PC_RAW + rbase + oa + j * q.HS * 16;
                rb[j] = (LDSQ char*)smem + PC_RAW + rbase + ob + j * q.HS * 16;
            }
        };
        auto x_desc = [&](const PcGeo& q) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + (size_t)q.n_img * img_bytes), 0, (int)(img_bytes * q.n_here), 0x00020000);
        };
        const float cb = xi == 1 ? 1.f : -1.f;
        const f32x2 cb2 = {cb, cb};
        const f32x2 k4 = {4.f, 4.f}, km5 = {-5.f, -5.f}, km4 = {-4.f, -4.f}, k2 = {2.f, 2.f}, km2 = {-2.f, -2.f};

        __amdgpu_buffer_rsrc_t rsrcL;
        int gL = 0;                             // next group (of its tile) the load stage requests
        // The request stage runs two groups (eight chunks) ahead of the store stage: with n = 8 chunks per tile (the C = 64 layers) it
        // works one whole tile ahead of the others (LEAD = 1), otherwise it moves to the next tile eight intervals before the end.
        const int LEAD = n == 8 ? 1 : 0;
        const int i_setup = n == 8 ? 6 : n - 9; // the interval in whose burst window the request stage's next address set is computed
        {
            const PcGeo q0 = pc_geo(p, tile0);
            setup_patch(q0, lp);
#pragma unroll
            for (int r = 0; r < NP; ++r) lpn[r] = lp[r];        // the fast path above leaves lpn alone
            setup_read(q0);
            rsrcL = x_desc(q0);
        }
        f32x4 rq[2][NP];                        // the pieces of two groups (group parity)
        f32x4 xa[3], xb[3], ya[3], yb[3], va[6];
        auto load_group = [&](f32x4 (&dst)[NP]) {      // request group gL of the load stage's tile
            const int so = min(gL, (n >> 2) - 1) * 128;     // (never past a tile's last group)
#pragma unroll
            for (int r = 0; r < NP; ++r)
                dst[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcL, goff[r], so, 0));
            ++gL;
        };
        // chunk (cq = 0..3 of its group) -> LDS raw[buf]: the sixteen lanes per instruction that hold this chunk's pieces store them
        // (exec mask by immediate; scalar + LDS instructions only -- a v_cmp in front of the stores would not issue before the consumer yields)
#define PC_ST1(i) "ds_write_b128 %" #i ", %[d" #i "] offset:%[off]\n\t"
#define PC_STORE12(MASK, src, buf)                                                                                                          \
        asm volatile("s_mov_b32 exec_lo, " MASK "\n\ts_mov_b32 exec_hi, " MASK "\n\t"                                                        \
                     PC_ST1(0) PC_ST1(1) PC_ST1(2) PC_ST1(3) PC_ST1(4) PC_ST1(5) PC_ST1(6) PC_ST1(7) PC_ST1(8) PC_ST1(9) PC_ST1(10) PC_ST1(11)  \
                     "s_mov_b64 exec, -1"                                                                                                    \
                     :: "v"(lp[0]), "v"(lp[1]), "v"(lp[2]), "v"(lp[3]), "v"(lp[4]), "v"(lp[5]), "v"(lp[6]), "v"(lp[7]), "v"(lp[8]), "v"(lp[9]),  \
                        "v"(lp[10]), "v"(lp[11]), [d0] "v"(src[0]), [d1] "v"(src[1]), [d2] "v"(src[2]), [d3] "v"(src[3]), [d4] "v"(src[4]),    \
                        [d5] "v"(src[5]), [d6] "v"(src[6]), [d7] "v"(src[7]), [d8] "v"(src[8]), [d9] "v"(src[9]), [d10] "v"(src[10]),          \
                        [d11] "v"(src[11]), [off] "n"((buf) * RAWB) : "memory")
        auto store_chunk = [&](const f32x4 (&src)[NP], const int cq, const int buf) {
            if (cq == 0) { if (buf) PC_STORE12("0x03030303", src, 1); else PC_STORE12("0x03030303", src, 0); }
            if (cq == 1) { if (buf) PC_STORE12("0x0c0c0c0c", src, 1); else PC_STORE12("0x0c0c0c0c", src, 0); }
            if (cq == 2) { if (buf) PC_STORE12("0x30303030", src, 1); else PC_STORE12("0x30303030", src, 0); }
            if (cq == 3) { if (buf) PC_STORE12("0xc0c0c0c0", src, 1); else PC_STORE12("0xc0c0c0c0", src, 0); }
        };
        // The transform in three steps that the pipeline places apart: twelve patch reads (LDS), 36 packed ops (vector ALU: they
        // only run while the matrix pipe is idle, so they sit in front of the barrier the consumer is about to wait at), six
        // fragment stores (LDS again: behind the barrier, under the next chunk's MFMAs).
        auto tr_read = [&](int buf) {
            const int bo = buf * RAWB;
            xa[0] = *reinterpret_cast<const f32x4 LDSQ*>(ra[0] + bo);
            xb[0] = *reinterpret_cast<const f32x4 LDSQ*>(rb[0] + bo);
            xa[1] = *reinterpret_cast<const f32x4 LDSQ*>(ra[2] + bo);
            xb[1] = *reinterpret_cast<const f32x4 LDSQ*>(rb[2] + bo);
            xa[2] = *reinterpret_cast<const f32x4 LDSQ*>(ra[0] + bo + 16);
            xb[2] = *reinterpret_cast<const f32x4 LDSQ*>(rb[0] + bo + 16);
            ya[0] = *reinterpret_cast<const f32x4 LDSQ*>(ra[1] + bo);
            yb[0] = *reinterpret_cast<const f32x4 LDSQ*>(rb[1] + bo);
            ya[1] = *reinterpret_cast<const f32x4 LDSQ*>(ra[3] + bo);
            yb[1] = *reinterpret_cast<const f32x4 LDSQ*>(rb[3] + bo);
            ya[2] = *reinterpret_cast<const f32x4 LDSQ*>(ra[1] + bo + 16);
            yb[2] = *reinterpret_cast<const f32x4 LDSQ*>(rb[1] + bo + 16);
        };
        auto tr_math = [&]() {
            const f32x4 T0 = fma4s(cb2, xb[0], xa[0]), T2 = fma4s(cb2, xb[1], xa[1]), T4 = fma4s(cb2, xb[2], xa[2]);
            const f32x4 T1 = fma4s(cb2, yb[0], ya[0]), T3 = fma4s(cb2, yb[1], ya[1]), T5 = fma4s(cb2, yb[2], ya[2]);
            va[0] = fma4s(km5, T2, fma4s(k4, T0, T4));
            va[5] = fma4s(km5, T3, fma4s(k4, T1, T5));
            {
                const f32x4 a = fma4s(km4, T2, T4), bq = fma4s(km4, T1, T3);
                va[1] = add4(a, bq);
                va[2] = sub4(a, bq);
            }
            {
                const f32x4 c = sub4(T4, T2), d = sub4(T3, T1);
                va[3] = fma4s(k2, d, c);
                va[4] = fma4s(km2, d, c);
            }
        };
        LDSQ char* const vrow0 = (LDSQ char*)smem + vlane;
        auto tr_write = [&](int buf) {
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<f32x4 LDSQ*>(vrow0 + buf * PC_VB + nu * 1024) = va[nu];
        };
#define PC_PIN_VA() asm volatile("" ::"v"(va[0]), "v"(va[1]), "v"(va[2]), "v"(va[3]), "v"(va[4]), "v"(va[5]))
        // ---- prologue of the block's first tile: V(0) in LDS, V(1) in registers, raw(2) in LDS, chunk 3 in registers, group 1 in flight
        load_group(rq[0]);
        load_group(rq[1]);
        if (LEAD) {                             // n = 8: both groups of the first tile are requested; the stage moves on to the block's second tile
            if (ntiles > 1) {
                const PcGeo q1 = pc_geo(p, tile0 + S);
                setup_patch(q1, lpn);
                rsrcL = x_desc(q1);
            } else {
#pragma unroll
                for (int r = 0; r < NP; ++r) goff[r] = kOob;
            }
            gL = 0;
        }
        store_chunk(rq[0], 0, 0);
        store_chunk(rq[0], 1, 1);
        PC_BAR();                               // P1: raw(0), raw(1) visible (the patch is shared by the four producer waves)
        tr_read(0);
        tr_math();
        tr_write(0);
        tr_read(1);
        PC_BAR();                               // P2: every wave has read raw[0] and raw[1]
        tr_math();                              // va = V(1)
        PC_PIN_VA();
        store_chunk(rq[0], 2, 0);               // raw(2)
        PC_BAR();                               // P3: V(0), raw(2) visible

        // ---- interval i of a tile (i = 0 .. n - 1), closed by the barrier inside the consumers' chunk i:
        //        fragments of chunk i + 1: registers -> V[(i + 1) & 1];   patch of chunk i + 3: registers -> raw[(i + 3) & 1];
        //        every fourth interval (chunk i + 3 was its group's last): request group (i + 3) / 4 + 2 into the freed registers;
        //        patch of chunk i + 2: raw[i & 1] -> registers, then its 36 packed ops; barrier.
        //      Chunks >= n of a tile ARE the next tile's first chunks: the request stage moves to the next tile after interval
        //      n - 12 (its addresses are recomputed in the burst window of interval n - 9), the store stage after n - 4, the read
        //      stage after n - 3 -- each piece of address state right behind its last use, so nothing is selected at run time.
        //      The tile's epilogue (second halves) follows its last interval; the next tile's interval 0 starts behind it.
        auto finish_prev = [&](const int tile_prev) {   // a finished tile's epilogue, second halves (barriers E1, E2, E3)
            const PcGeo qp = pc_geo(p, tile_prev);
            PcEpi epi;
            pc_epi_vectors(p, qp, tid, epi);    // requested here, used behind the barrier
            PC_TR(30);
            PC_BAR();                           // E1: ex holds n-tile 0
            PC_TR(31);
            if (!(SEAM_W24PC_ABL & 16)) pc_finish(p, qp, ex, 0, tid, false, epi);
            PC_TR(32);
            PC_BAR();                           // E2: ex is free again
            PC_TR(33);
            PC_BAR();                           // E3: ex holds n-tile 1
            PC_TR(34);
            if (!(SEAM_W24PC_ABL & 16)) pc_finish(p, qp, ex, 1, tid, false, epi);
            PC_TR(35);
        };
        int tile = tile0;
        for (int k = 0; k < ntiles; ++k) {
            const bool has_next = k + 1 < ntiles, has_lead = k + 1 + LEAD < ntiles;
            for (int i0 = 0; i0 < n; i0 += 8) {
                // interval i = i0 + j; i0 is a multiple of 8: every index below is a compile-time constant of j (a template argument,
                // not an unrolled loop variable: the register sets must be split into registers before any loop pass runs)
                auto ivl = [&](auto jt) {
                    constexpr int j = decltype(jt)::value;
                    PC_TR(1);
#ifdef SEAM_W24PC_PSLEEP
                    __builtin_amdgcn_s_sleep(SEAM_W24PC_PSLEEP);
#endif
                    tr_write((j + 1) & 1);
#if SEAM_W24PC_ABL & 256
                    {       // this wave pair's twelve fragments of a chunk two ahead: global -> LDS, no registers (never waited for: timing only)
                        const int nb = n * 24576;
                        const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(
                            (void*)((const char*)p.u + (size_t)pc_geo(p, tile).tn * NT * nb), 0, NT * nb, 0x00020000);
                        const int ck = min(i0 + j + 2, n - 1) * 24576;
#pragma unroll
                        for (int r = 0; r < 12; ++r)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (LDSQ void*)((LDSQ char*)smem + PC_EX + (xi * 12 + r) * 1024), 16,
                                                                     (xi * 6 * 64 + lane) * 16 + (r % 6 & 3) * 1024, ck + ((r % 6) >> 2) * 4096 + (r / 6) * nb, 0, 0);
                    }
#endif
#ifdef SEAM_W24PC_PSLEEP
                    __builtin_amdgcn_s_sleep(SEAM_W24PC_PSLEEP);
#endif
                    if (!(SEAM_W24PC_ABL & 1)) {
                        if (!(SEAM_W24PC_ABL & 64)) store_chunk(rq[((j + 3) >> 2) & 1], (j + 3) & 3, (j + 3) & 1);     // 64: requests only
                        if ((j & 3) == 0 && !(SEAM_W24PC_ABL & 128)) load_group(rq[((j + 3) >> 2) & 1]);           // 128: LDS stores only
                    }
                    tr_read(j & 1);
                    PC_TR(2);
                    if (!(SEAM_W24PC_ABL & 8)) tr_math();
                    PC_PIN_VA();
                    // address state for the next tile, recomputed in this burst window right behind its last use
                    if ((j == 7 || j == 6) && i0 + j == i_setup) {      // the request stage (n >= 16: its last request for this tile was at i = n - 12)
                        if (has_lead) {
                            const PcGeo qn = pc_geo(p, tile + (1 + LEAD) * S);
                            setup_patch(qn, lpn);
                            rsrcL = x_desc(qn);
                        } else {                            // no next tile: the requests fall outside every image
#pragma unroll
                            for (int r = 0; r < NP; ++r) goff[r] = kOob;
                        }
                        gL = 0;
                    }
                    if (j == 4 && i0 == n - 8 && has_next) {                                    // i = n - 4: the store stage
#pragma unroll
                        for (int r = 0; r < NP; ++r) lp[r] = lpn[r];
                    }
                    if (j == 5 && i0 == n - 8 && has_next) setup_read(pc_geo(p, tile + S));     // i = n - 3
                    PC_TR(6);
                    // the fragments are inputs of the barrier statement: hipcc otherwise sinks the packed ops below it
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::"v"(va[0]), "v"(va[1]), "v"(va[2]), "v"(va[3]), "v"(va[4]), "v"(va[5]) : "memory");
                    SB();
                };
                ivl(std::integral_constant<int, 0>{});
                ivl(std::integral_constant<int, 1>{});
                ivl(std::integral_constant<int, 2>{});
                ivl(std::integral_constant<int, 3>{});
                ivl(std::integral_constant<int, 4>{});
                ivl(std::integral_constant<int, 5>{});
                ivl(std::integral_constant<int, 6>{});
                ivl(std::integral_constant<int, 7>{});
            }
            finish_prev(tile);                  // this tile's epilogue (the consumers are past their last chunk's barrier)
            tile += S;
        }
#undef PC_PIN_VA
#undef PC_STORE12
#undef PC_ST1
    } else {
        // =================================================== consumer ===================================================
        f32x16 acc[6][NT];
        const int ntile_bytes = n * 24576;
        int uoff0 = (xi * 6 * 64 + lane) * 16;
        f32x4 bq[RING][NT];                     // B fragments: slot j % RING holds position instance j = 6 * chunk + nu
        constexpr int AD = SEAM_W24PC_AD;       // A fragments are requested AD positions ahead of their MFMAs
        f32x4 aq[AD + 1];                       // A fragments: position nu in aq[nu % (AD + 1)]
        static_assert(6 % (AD + 1) == 0 && AD >= 1 && AD <= 2, "the fragment ring's phase repeats every chunk");
        auto read_a = [&](int buf, int nu) -> f32x4 { return *reinterpret_cast<const f32x4*>(smem + vlane + buf * PC_VB + nu * 1024); };
        PC_BAR();                               // P1
        PC_BAR();                               // P2
        PC_BAR();                               // P3
        int tile = tile0;
        auto u_desc = [&](const int tn) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.u + (size_t)tn * NT * ntile_bytes), 0, NT * ntile_bytes, 0x00020000);
        };
        __amdgpu_buffer_rsrc_t u_rsrc = u_desc(pc_geo(p, tile0).tn);
        for (int k = 0; k < ntiles; ++k) {
            const PcGeo q = pc_geo(p, tile);
            auto load_b = [&](int slot_, int nu, int chunk) {
                const int so = min(chunk, n - 1) * 24576 + (nu >> 2) * 4096;     // (chunks past the end: the last one again)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#if SEAM_W24PC_ABL & 256
                    (void)so;
                    bq[slot_][nt] = *reinterpret_cast<const f32x4*>(smem + PC_EX + ((xi * 6 + nu) * 2 + nt) * 1024 + lane * 16);
#else
                    bq[slot_][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, uoff0 + (nu & 3) * 1024, so + nt * ntile_bytes, 0));
#endif
                    SB();
                }
            };
            auto ring_preload = [&]() {         // the tile's first RING position instances, in ring order, pinned: the K loop's vmcnt
#pragma unroll                                  // waits count on that order
                for (int j = 0; j < RING; ++j) {
                    SB();
                    load_b(j, j % 6, j / 6);
                }
                SB();
            };
            if (k == 0) ring_preload();         // later tiles: requested in the previous tile's epilogue
#pragma unroll
            for (int a = 0; a < AD; ++a) aq[a] = read_a(0, a);
            // one chunk: positions 0..5 in order; c = chunk parity (V buffer), t = chunk index; FIRST: the tile's first chunk
            // multiplies its first k-step into a zero constant instead of clearing 192 accumulators
            auto chunk = [&](const int t, const int c, const bool first) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int jl = 6 * c + i, sl = jl % RING;
                    SB();
#ifdef SEAM_W24PC_TRACE_POS
                    PC_TR(10 + i);
#endif
                    // the fragment AD positions ahead (the next chunk's first ones come from the other V buffer, behind the barrier)
                    aq[(i + AD) % (AD + 1)] = i + AD < 6 ? read_a(c, i + AD) : read_a(c ^ 1, i + AD - 6);
                    SB();
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if (first && kk == 0) {
                                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[sl][nt][kk], aq[i % (AD + 1)][kk], z, 0, 0, 0);
                            } else {
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[sl][nt][kk], aq[i % (AD + 1)][kk], acc[i][nt], 0, 0, 0);
                            }
                            SB();
                        }
                    if (!(SEAM_W24PC_ABL & 2)) {
                        const int j2 = jl + RING;                   // the instance that takes over this slot (chunks past the end
                        load_b(sl, j2 % 6, t - c + j2 / 6);         // re-read the last chunk, never used)
                    }
                    SB();
                    if (i == 5 - AD) {          // every fragment of this chunk is in registers (or on its way, waited for by the barrier
                        PC_TR(8);               // statement): the producers may overwrite V[c]
                        PC_BAR();
                        PC_TR(9);
                    }
                }
            };
            static_assert(12 % RING == 0, "the ring's phase repeats every two chunks");
            chunk(0, 0, true);
            chunk(1, 1, false);
            for (int t = 2; t < n; t += 2) {
                asm volatile("" : "+v"(uoff0));
                chunk(t, 0, false);
                chunk(t + 1, 1, false);         // nchunks is even (wino24_pc): one straight two-chunk body, exact vmcnt waits
            }
            // ---- epilogue: the nu half of the output transform (A4t: 6 -> 4) on register pairs into the exchange array ----
            //   columns (A4t over nu): Y0 = m0+m1+m2+m3+m4, Y1 = (m1-m2) + 2(m3-m4), Y2 = (m1+m2) + 4(m3+m4), Y3 = (m1-m2) + 8(m3-m4) + m5
            //   rows (A2t over xi, across the waves, in pc_finish): y0 = q0+q1+q2, y1 = q1-q2-q3
            PcEpi epi;
            pc_epi_vectors(p, q, tid, epi);     // requested here, used behind the first epilogue barrier
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                PC_TR(20 + 4 * nt);
                if (nt == 1) PC_BAR();          // E2: the finishers of n-tile 0 are done with ex
                PC_TR(21 + 4 * nt);
                if (!(SEAM_W24PC_ABL & 16)) {
                    const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c8 = {8.f, 8.f};
                    // The MFMAs ran with the operand roles swapped (rows = channels, columns = tiles; the products and their order are
                    // the same, so are the results): a lane holds ONE tile (lane & 31) and, in registers 4m..4m+3, the four consecutive
                    // channels 8m + 4 (lane >> 5) + 0..3 -- the exchange rows [tile][32 channels] take 16-byte stores (16 per n-tile
                    // instead of 64 4-byte ones).  16-byte channel group g of a row sits at slot g ^ (tile & 7): conflict-free for the
                    // stores (eight consecutive tiles per LDS phase) and for pc_finish's reads (eight groups of one row).
                    const int tl = lane & 31;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        f32x2 yy[4][2];
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int h = 2 * m + hh;
                            auto rd = [&](int nu) -> f32x2 { return f32x2{acc[nu][nt][2 * h], acc[nu][nt][2 * h + 1]}; };
                            const f32x2 m0 = rd(0), m1 = rd(1), m2 = rd(2), m3 = rd(3), m4 = rd(4), m5 = rd(5);
                            const f32x2 s12 = pk_add(m1, m2), d12 = pk_sub(m1, m2), s34 = pk_add(m3, m4), d34 = pk_sub(m3, m4);
                            yy[0][hh] = pk_add(pk_add(m0, s12), s34);
                            yy[1][hh] = pk_fma_s(c2, d34, d12);
                            yy[2][hh] = pk_fma_s(c4, s34, s12);
                            yy[3][hh] = pk_add(pk_fma_s(c8, d34, d12), m5);
                        }
                        const int grp = (2 * m + (lane >> 5)) ^ (tl & 7);
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            *reinterpret_cast<f32x4*>(&ex[((xi * 4 + b) * 32 + tl) * 32 + grp * 4]) =
                                __builtin_shufflevector(yy[b][0], yy[b][1], 0, 1, 2, 3);
                    }
                } else if (nt == 1) {
                    float sum = 0.f;
#pragma unroll
                    for (int nu = 0; nu < 6; ++nu)
#pragma unroll
                        for (int m = 0; m < NT; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) sum += acc[nu][m][r];
                    if (sum == 123456.789f) p.y[tid] = sum;
                }
                if (nt == NT - 1 && k + 1 < ntiles) {   // the accumulators are dead: the next tile's first weight fragments travel under
                    u_rsrc = u_desc(pc_geo(p, tile + S).tn);    // the rest of the epilogue
                    ring_preload();
                }
                PC_TR(22 + 4 * nt);
                PC_BAR();                       // E1 / E3: ex holds this n-tile
                PC_TR(23 + 4 * nt);
                if (!(SEAM_W24PC_ABL & 16)) pc_finish(p, q, ex, nt, tid, true, epi);
            }
            tile += S;
        }
#ifdef SEAM_W24PC_TRACE
        if (p.trace && tid == 0) {      // per-block span (cycles) and tile count, behind the two traced waves' stamp areas
            p.trace[8 * 4096 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - blk_t0;
            p.trace[8 * 4096 + 2 * blockIdx.x + 1] = (unsigned long long)ntiles;
        }
#endif
    }
#undef SB
}

// OIHW fp32 [K, Cin, 3, 3] -> U = G2 g G4t in MFMA fragment order: [K/32][Cstore/8][24][64][4]
//   element (tn, chunk, p = 6*xi + nu, lane, e) = U_p[n = 32*tn + (lane & 31)][c = 8*chunk + 4*(lane >> 5) + e]
// mode 0: forward weights; mode 2: input-gradient weights (taps rotated 180 degrees, channels swapped), as in
// seam_pack_conv_weight_f32.  fp64 arithmetic, one rounding.
__global__ void wino24_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int K, int Cin, int Cstore, int mode) {
    const int nch = Cstore / 8;
    const size_t total = (size_t)(K / 32) * nch * 64;
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
                const double a0 = gg[q][0], a1 = gg[q][1], a2 = gg[q][2];
                double uu[6];
                uu[0] = a0 / 4.0;
                uu[1] = -(a0 + a1 + a2) / 6.0;
                uu[2] = -(a0 - a1 + a2) / 6.0;
                uu[3] = a0 / 24.0 + a1 / 12.0 + a2 / 6.0;
                uu[4] = a0 / 24.0 - a1 / 12.0 + a2 / 6.0;
                uu[5] = a2;
                const size_t base = ((((size_t)tn * nch + chunk) * 24 + q * 6) * 64 + lane) * 4 + e;
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) out[base + (size_t)nu * 256] = (float)uu[nu];
            }
        }
    }
}

// ---- block layout (32 tile slots of 2 x 4 outputs per block) --------------------------------------------------------
struct Layout {
    int nreg, G, stack, PH;
    int rx0[3], ry0[3], rxe[3], rye[3], TX[3], TY[3], bx[3], by[3];
    long per_img, blocks;      // blocks: per n-tile, for N images
};

constexpr int CAP = 32;

// LDS entries per channel half for a patch of tx tiles x ph rows (the kernel's HS / PR / NENT)
inline int lds_entries(int tx, int ph) {
    const int hs = (tx + 1) | 1, pr = 8 * hs + (tx & 7), n0 = pr * ((ph + 1) / 2);
    return n0 + ((4 - n0) & 7);
}

inline bool patch_ok(int tx, int ty) {
    return tx >= 1 && ty >= 1 && tx * ty <= CAP && (4 * tx + 2) * (2 * ty + 2) <= NPIXMAX && lds_entries(tx, 2 * ty + 2) <= ENTMAX;
}

inline long best_uniform(int w, int h, int& TX, int& TY) {
    long best = -1, best_cost = -1;
    for (int ty = 1; ty <= CAP && ty <= h; ++ty)
        for (int tx = 1; tx * ty <= CAP && tx <= w; ++tx) {
            if (!patch_ok(tx, ty)) continue;
            const long nb = (long)((w + tx - 1) / tx) * ((h + ty - 1) / ty);
            const long cost = nb * 4096 + (4 * tx + 2) * (2 * ty + 2);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nb; TX = tx; TY = ty; }
        }
    return best;
}

inline Layout choose_layout(int N, int tiles_x, int tiles_y, size_t in_img_bytes, size_t out_img_bytes) {
    Layout L;
    L.nreg = 1; L.G = 1; L.stack = 0; L.PH = 0;
    Layout S;
    S.blocks = -1;
    if (tiles_x <= CAP) {      // stacked candidate: TY consecutive tile rows of the batch at full width per block
        const int t = tiles_y, pitch = 2 * t + 2, pw = 4 * tiles_x + 2;
        for (int ty = CAP / tiles_x; ty >= 1; --ty) {
            int phmax = 0, gmax = 0;
            for (int b = 0; b < t; ++b) {
                const int tyf = (int)(((long)b * ty) % t);
                const int last = tyf + ty - 1;
                const int gl = last / t, tyl = last - gl * t;
                const int span = pitch * gl + 2 * tyl + 4 - 2 * tyf;
                if (span > phmax) phmax = span;
                if (gl + 1 > gmax) gmax = gl + 1;
            }
            if (phmax * pw > NPIXMAX || lds_entries(tiles_x, phmax) > ENTMAX) continue;
            if ((size_t)gmax * in_img_bytes >= kOob || (size_t)gmax * out_img_bytes >= kOob) continue;
            S.nreg = 1; S.stack = 1; S.PH = phmax; S.G = gmax;
            S.rx0[0] = S.ry0[0] = 0; S.rxe[0] = tiles_x; S.rye[0] = tiles_y; S.TX[0] = tiles_x; S.TY[0] = ty; S.bx[0] = S.by[0] = 1;
            S.per_img = 1;
            S.blocks = ((long)N * t + ty - 1) / ty;
            break;
        }
    }
    long best_blocks = best_uniform(tiles_x, tiles_y, L.TX[0], L.TY[0]);
    L.rx0[0] = L.ry0[0] = 0; L.rxe[0] = tiles_x; L.rye[0] = tiles_y;
    L.bx[0] = (tiles_x + L.TX[0] - 1) / L.TX[0]; L.by[0] = (tiles_y + L.TY[0] - 1) / L.TY[0];
    for (int ty = 1; ty <= CAP && ty <= tiles_y; ++ty)
        for (int tx = 1; tx * ty <= CAP && tx <= tiles_x; ++tx) {
            if (!patch_ok(tx, ty)) continue;
            const int mx = tiles_x / tx, my = tiles_y / ty;
            const int wm = mx * tx, hm = my * ty;
            if (wm == 0 || hm == 0) continue;
            Layout C;
            C.G = 1; C.nreg = 1; C.stack = 0; C.PH = 0;
            C.rx0[0] = 0; C.ry0[0] = 0; C.rxe[0] = wm; C.rye[0] = hm; C.TX[0] = tx; C.TY[0] = ty; C.bx[0] = mx; C.by[0] = my;
            long nb = (long)mx * my;
            if (wm < tiles_x) {
                const int r = C.nreg++;
                const long b = best_uniform(tiles_x - wm, tiles_y, C.TX[r], C.TY[r]);
                C.rx0[r] = wm; C.ry0[r] = 0; C.rxe[r] = tiles_x; C.rye[r] = tiles_y;
                C.bx[r] = (tiles_x - wm + C.TX[r] - 1) / C.TX[r]; C.by[r] = (tiles_y + C.TY[r] - 1) / C.TY[r];
                nb += b;
            }
            if (hm < tiles_y) {
                const int r = C.nreg++;
                const long b = best_uniform(wm, tiles_y - hm, C.TX[r], C.TY[r]);
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

// n-tiles per block.  NT = 2 (one block per CU, 192 accumulators in AccVGPRs, every A fragment and the raw patch feeding twice
// the MFMAs) is built and parity-clean but NOT the default: as compiled by hipcc 7.2 it ties on the 200^2 layers and loses
// 3-13 % elsewhere -- the register allocator parks the loop-invariant LDS / load addresses and 8 transform registers in spare
// AccVGPRs and re-reads them (~60 v_accvgpr_read per 96 MFMAs) although ~90 arch VGPRs stay unused in the loop, and an asm MFMA
// with pinned register classes did not change that.  SEAM_W24_NT=2 enables it for experiments.
inline int wino24_nt(int K, int C, long blocks_nt1) {
    // NT = 2 pays where the K loop is long enough to amortise a block that is alone on its CU (C >= 256) and the launch still fills
    // the chip several times over with half as many blocks (measured per layer shape with tools/w24_ab.py; both forms compute
    // bit-identical results, so the choice may depend on the batch).  SEAM_W24_NT=1|2 forces a form.
    const int force = seam_opt::get(seam_opt::W24_NT);
    if (K % 64) return 1;
    if (force == 1 || force == 2) return force;
    // round 5: on the producer / consumer kernel the shorter K loops of the C = 64 / 128 layers pay as well (80 x 100^2 x 128: -6 %)
    const int pc = seam_opt::get(seam_opt::W24_PC);
    const int cmin = pc && C % 64 == 0 ? 64 : 256;
    return (C >= cmin && blocks_nt1 / 2 >= 1024) ? 2 : 1;
}

// XCD groups the n-tiles are split over (1 = every XCD walks all n-tiles of its patches).  Default off until measured per shape.
inline int wino24_nsplit(int C, int K, long patch_blocks) {
    (void)C; (void)K; (void)patch_blocks;
    return 1;
}

// the producer / consumer kernel takes the NT = 2 launches (SEAM_W24_PC=0: conv3x3_wino24<2>, the round-4 kernel, stays selectable)
inline bool wino24_pc(const Wino24Args& a) {
    const int want = seam_opt::get(seam_opt::W24_PC);
    return want && a.nt == 2 && a.nsplit == 1 && a.nchunks >= 8 && a.nchunks % 8 == 0;
}

int wino24_plan(Wino24Args& a, int N, int H, int W, int C, int K, int pad, long& blocks) {
    if (!wino_ok(C, K, 3, 3, 1) || N <= 0) return (int)hipErrorInvalidValue;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K;
    a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2; a.pad = pad;
    if (a.Ho <= 0 || a.Wo <= 0) return (int)hipErrorInvalidValue;
    if ((size_t)H * W * C * 4 >= kOob || (size_t)a.Ho * a.Wo * K * 4 >= kOob) return (int)hipErrorInvalidValue;
    const int tiles_x = (a.Wo + 3) / 4, tiles_y = (a.Ho + 1) / 2;
    const Layout pp = choose_layout(N, tiles_x, tiles_y, (size_t)H * W * C * 4, (size_t)a.Ho * a.Wo * K * 4);
    a.nt = wino24_nt(K, C, pp.blocks * (K / 32));
    a.tiles_n = K / (32 * a.nt);
    a.nchunks = C / 8;
    a.nreg = pp.nreg; a.G = pp.G; a.per_img = (int)pp.per_img;
    a.stack = pp.stack; a.PH = pp.PH; a.tiles_y = tiles_y;
    for (int r = 0; r < 3; ++r) {
        const int q = r < pp.nreg ? r : 0;
        a.rx0[r] = pp.rx0[q]; a.ry0[r] = pp.ry0[q]; a.rxe[r] = pp.rxe[q]; a.rye[r] = pp.rye[q];
        a.TX[r] = pp.TX[q]; a.TY[r] = pp.TY[q]; a.bx[r] = pp.bx[q]; a.by[r] = pp.by[q];
    }
    blocks = pp.blocks * a.tiles_n;
    {
        const int want = seam_opt::get(seam_opt::W24_NSPLIT);
        int ns = want > 0 ? want : wino24_nsplit(C, K, pp.blocks);
        while (ns > 1 && (a.tiles_n % ns || 8 % ns)) ns >>= 1;
        a.nsplit = ns < 1 ? 1 : ns;
        a.tns = a.tiles_n / a.nsplit;
        const int parts = 8 / a.nsplit;
        a.part_q = (int)(pp.blocks / parts); a.part_r = (int)(pp.blocks % parts);
        if (a.nsplit > 1) blocks = 8L * (a.part_q + (a.part_r ? 1 : 0)) * a.tns;
    }
    if (blocks >= (1L << 24)) return (int)hipErrorInvalidValue;     // also keeps every fdiv operand inside a * d < 2^32
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
    a.m_tns = magic(a.tns);
    a.m_tiles_n = magic(a.tiles_n); a.m_per_img = magic(a.per_img); a.m_tys = magic(tiles_y); a.m_pitch = magic(2 * tiles_y + 2);
    for (int r = 0; r < 3; ++r) { a.m_bx[r] = magic(a.bx[r]); a.m_TX[r] = magic(a.TX[r]); a.m_PW[r] = magic(4 * a.TX[r] + 2); }
    return 0;
}

}  // namespace

extern "C" {

long long seam_wino24_weight_floats(int K, int Cstore) { return (long long)K * Cstore * 24; }

int seam_pack_conv_weight_wino24_f32(const float* w, float* u_packed, int K, int Cin, int Cstore, int mode, void* stream) {
    if (!wino_ok(Cstore, K, 3, 3, 1) || Cin > Cstore) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)(K / 32) * (Cstore / 8) * 64;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(wino24_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, u_packed, K, Cin, Cstore, mode);
    return (int)hipGetLastError();
}

/* MFMA issues of the launch in units of 32x32x8-channel position GEMMs (blocks x 32 tile slots x 24 positions; 0 = unsupported):
 * the figure ops.conv2d compares with seam_wino_issue_slots to pick the cheaper Winograd form for a layer shape. */
long long seam_wino24_issue_slots(int N, int H, int W, int C, int K, int pad) {
    Wino24Args a;
    long blocks;
    if (wino24_plan(a, N, H, W, C, K, pad, blocks)) return 0;
    const long work = a.nsplit > 1 ? ((long)(8 / a.nsplit) * a.part_q + a.part_r) * a.tiles_n : blocks;     // without the padding blocks
    return (long long)work * a.nt * 32 * 24;
}

/* 1 when the launcher runs this layer shape on the producer / consumer kernel conv3x3_wino24pc (NT = 2 work split over two wave
 * groups), 0 for conv3x3_wino24<NT>; -1 = unsupported shape */
int seam_wino24_form(int N, int H, int W, int C, int K, int pad) {
    Wino24Args a;
    long blocks;
    if (wino24_plan(a, N, H, W, C, K, pad, blocks)) return -1;
    return wino24_pc(a) ? 1 : 0;
}

/* n-tiles per block (the kernel's template argument NT = 1 | 2) the launcher picks for this layer shape; 0 = unsupported */
int seam_wino24_variant(int N, int H, int W, int C, int K, int pad) {
    Wino24Args a;
    long blocks;
    if (wino24_plan(a, N, H, W, C, K, pad, blocks)) return 0;
    return a.nt;
}

int seam_conv3x3_wino24_f32(const float* x, const float* u_packed, const float* scale, const float* shift, const float* residual,
                            float* y, int N, int H, int W, int C, int K, int pad, int relu, void* stream) {
    Wino24Args a;
    long blocks;
    const int rc = wino24_plan(a, N, H, W, C, K, pad, blocks);
    if (rc) return rc;
    a.x = x; a.u = u_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.relu = relu;
    a.trace = nullptr;
    a.total_tiles = 0;
    const int dyn = seam_opt::get(seam_opt::W24_DYNLDS);     // dev knob: occupancy experiments
    if (wino24_pc(a)) {
        // > 64 KiB of dynamic LDS needs the attribute once per device (an atomic flag per device: the ABI is thread-safe per stream)
        static std::atomic<unsigned> attr_done{0};
        static std::atomic<int> cus[32];
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (!(attr_done.load(std::memory_order_acquire) & bit)) {
            const hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wino24pc<SEAM_W24PC_RING>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
            if (e != hipSuccess) return (int)e;
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
            cus[dev & 31].store(ncu, std::memory_order_relaxed);
            attr_done.fetch_or(bit, std::memory_order_release);
        }
        // persistent grid: one block per CU walks its XCD's tile range (SEAM_W24_PERSIST=0: one tile per block)
        const int persist = seam_opt::get(seam_opt::W24_PERSIST);
        a.total_tiles = (int)blocks;
        for (int r = 0; r < 3; ++r) {
            a.rg[r].TX = a.TX[r]; a.rg[r].TY = a.TY[r]; a.rg[r].bx = a.bx[r]; a.rg[r].by = a.by[r];
            a.rg[r].rx0 = a.rx0[r]; a.rg[r].ry0 = a.ry0[r]; a.rg[r].rxe = a.rxe[r]; a.rg[r].rye = a.rye[r];
            a.rg[r].m_bx = a.m_bx[r]; a.rg[r].m_TX = a.m_TX[r]; a.rg[r].m_PW = a.m_PW[r];
            for (int e = 0; e < 5; ++e) a.rg[r].pad_[e] = 0;
        }
        const int ncu = cus[dev & 31].load(std::memory_order_relaxed);
        const unsigned grid = (unsigned)(persist && blocks > ncu ? ncu : blocks);
#ifdef SEAM_W24PC_TRACE
        static seam_dev::TraceBuf tb;
        a.trace = seam_dev::trace_begin(tb, 8 * 4096 + 2 * 1024);
#endif
        hipLaunchKernelGGL(conv3x3_wino24pc<SEAM_W24PC_RING>, dim3(grid), dim3(512), PC_LDS, (hipStream_t)stream, a);
#ifdef SEAM_W24PC_TRACE
        seam_dev::trace_end(tb, 3, 4096, true, grid < 1024 ? grid : 1024);
#endif
        return (int)hipGetLastError();
    }
    if (a.nt == 2) hipLaunchKernelGGL(conv3x3_wino24<2>, dim3((unsigned)blocks), dim3(256), dyn, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv3x3_wino24<1>, dim3((unsigned)blocks), dim3(256), dyn, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // extern "C"
