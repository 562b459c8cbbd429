// seam_conv.hip -- implicit-GEMM convolution on gfx950 matrix cores: exact fp32 (the default path)
// and fp16-in / fp32-accumulate (the path BASELINE config 5 names).
//
// One kernel family serves every dense contraction of the path (ResNet-50 body, FPN, RPN head,
// box/mask heads, the match trunk's valid 3x3 convs and its Linear) -- see include/seam_hip.h.
//
// GEMM view (TN):  Y[M, K] = A[M, kred] * B[K, kred]^T
//   M    = N*Ho*Wo output pixels, row m -> (n, ho, wo)          (NHWC output == row-major [M][K])
//   kred = reduction index, walked in 128-BYTE chunks (32 floats / 64 halves) ordered
//          (r, c-chunk, s) [C >= one chunk], so the horizontal taps of one (row, channel-chunk) are
//          consecutive chunks; A is gathered on the fly from the NHWC input (hardware zero fill for
//          padding / tails); B = pre-packed weights, TILE-CONTIGUOUS: [n_tile][chunk][BN][128 B]
//          (one 16 KiB slab per chunk: no power-of-two row stride, 4 TLB pages).
// Tiling for CDNA4 (wave64, 4 SIMDs/CU):
//   block 256 threads = 4 waves (2x2); block tile BM x BN x (128 B of k); wave tile (BM/2) x (BN/2)
//   of 32x32 MFMA tiles (16 accumulator VGPRs each):
//     fp32: v_mfma_f32_32x32x2_f32  (64 cyc/issue = the fp32 rate; bit-exact fp32 fma chain)
//     fp16: v_mfma_f32_32x32x16_f16 (fp32 accumulate)
//   Both operand types share ONE byte layout: an LDS row is 128 B of k + 16 B pad (9 x 16-B slots,
//   odd => ds_read_b128 / ds_write_b128 lane groups are conflict-free); lane l reads 16 B of row
//   (l&31) at k-slot (l>>5): 4 floats feed 4 fp32 MFMAs (k-pairs {j, j+4}, same permutation on A
//   and B), 8 halves feed 1 fp16 MFMA.
//   Pipeline: raw buffer loads (SGPR descriptor, 32-bit lane offsets, hardware OOB zero fill) ->
//   registers -> LDS, double-buffered LDS, TWO register sets (loads issued during chunk t are for
//   chunk t+2) with counted vmcnt, fragments double-buffered ACROSS the chunk barrier, every
//   load / LDS write unconditional (a branch around a load makes hipcc serialise it behind
//   vmcnt(0)).  Epilogue: scale/shift (bias / folded BN) + residual + ReLU, branch-free buffer ops.
//   blockIdx -> tile mapping is XCD-aware (consecutive tiles stay on one XCD's L2; bijective form).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "seam_opts.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CHUNK_BYTES = 128;                    // k bytes per row per chunk
constexpr int LDB = CHUNK_BYTES + 16;               // padded LDS row (bytes)
constexpr unsigned kOob = 0x80000000u;

struct ConvArgs {
    const void* x;
    const void* w;
    const float* scale;
    const float* shift;
    const void* res;
    void* y;
    int N, H, W, C;
    int Ho, Wo, K;
    int R, S, stride, pad;
    int kred;          // padded reduction length in elements (multiple of the chunk)
    int M;
    int relu;
    int y_f32;         // fp16 kernel only: write fp32 output (descriptor heads stay fp32)
    int vec_epi;       // fp16 kernel: 16-byte epilogue through the LDS transpose (dev knob SEAM_F16_VEC_EPILOGUE=0 turns it off)
    int slab_bn;       // rows per packed weight slab (128 or 64; fixed at pack time, >= the tile's BN)
    int tiles_m, tiles_n;
    // DUAL kernels (fp32): the reduction runs over TWO 1x1 sources -- chunks [0, C/32) from x [N,Ho,Wo,C] (stride 1) and the rest
    // from x2 [N,H2,W2,C2] sampled at (ho*stride2, wo*stride2); kred = C + C2.  ResNet downsample blocks: bn3(conv3(h)) +
    // bn_d(conv_d(x)) as ONE GEMM (scales folded into the weights), so the shortcut never goes to memory and back.
    const void* x2;
    int H2, W2, C2, stride2;
    // Row -> (image, ho, wo) without integer divisions (each costs ~20 VALU instructions, per row, per tile -- and VALU
    // instructions delay the matrix pipe): m_HoWo / m_Wo = ceil(2^32 / d), valid for the operand ranges of split_row() when
    // div_fast is set (Ho*Wo*Wo < 2^32; the host checks).  A tile's rows are first made local to its first image.
    unsigned m_HoWo, m_Wo;
    int div_fast;
    int epi_prio;      // 1: s_setprio 3 for the epilogue (dev knob SEAM_EPI_PRIO=0 turns it off)
    int rH, rW;        // > 0 (fp32, K % 4 == 0 only): `res` is a coarser map [N, rH, rW, K] added through a nearest-neighbour
                       // upsample to [Ho, Wo] (ATen: src = min(floor(dst * rH / Ho), rH - 1)) -- the FPN top-down merge
};

// rl = row index local to the tile's first image (0 <= rl < Ho*Wo + tile rows) -> images past that one, ho, wo
__device__ __forceinline__ void split_row(const ConvArgs& p, int HoWo, int rl, int& nl, int& ho, int& wo) {
    if (p.div_fast) {
        nl = HoWo >= 256 ? (rl >= HoWo ? 1 : 0) : (HoWo == 1 ? rl : (int)__umulhi((unsigned)rl, p.m_HoWo));
        const int rm = rl - nl * HoWo;
        ho = p.Wo == 1 ? rm : (int)__umulhi((unsigned)rm, p.m_Wo);
        wo = rm - ho * p.Wo;
    } else {
        nl = rl / HoWo;
        const int rm = rl - nl * HoWo;
        ho = rm / p.Wo;
        wo = rm - ho * p.Wo;
    }
}

// Measured and not kept (round 2, profiles/r02_f16_wave128.txt): the fp16 256 x 128 tile as 4 waves of 128 x 64 (0.75 LDS fragment
// reads per MFMA instead of 1, 128 accumulators, one block of 4 waves per CU) runs 763 TFLOP/s on the 80 x 200^2 x 256 -> 256 layer
// against 859 for the same tile as 8 waves of 64 x 64: the 16-bit kernels are bound by load latency, not by LDS bandwidth, and four
// waves per CU hide less of it.
template <typename T, int BM, int BN, int NW, bool DUAL = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_igemm(const ConvArgs p) {
    constexpr bool F16 = sizeof(T) == 2;
    constexpr int ES = (int)sizeof(T);
    constexpr int EPV = 16 / ES;                 // elements per 16-byte vector (4 / 8)
    constexpr int BKE = CHUNK_BYTES / ES;        // k elements per chunk (32 / 64)
    constexpr int WM = BM / (NW / 2), WN = BN / 2;      // wave tile: NW waves as (NW/2) x 2
    constexpr int MT = WM / 32, NT = WN / 32;    // 32x32 MFMA tiles per wave
    constexpr int RP = 8 * NW;                   // tile rows covered by one load pass of the block (8 lanes per row)
    constexpr int AI = BM / RP, BI = BN / RP;    // 16-byte loads per thread per chunk
    constexpr int Q = (F16 ? 1 : 4) * MT * NT;   // MFMAs per k-slot (32 B of k per row)

    __shared__ __attribute__((aligned(16))) char As[2][BM * LDB];
    __shared__ __attribute__((aligned(16))) char Bs[2][BN * LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm0 = (wid >> 1) * WM;
    const int wn0 = (wid & 1) * WN;

    // ---- persistent tiles --------------------------------------------------------------------
    // A block walks tiles vb = blockIdx.x, + gridDim.x, ... (grid = the resident block slots, a multiple of 8, so a block
    // stays on its XCD); vb -> tile is the XCD-aware bijection over all tiles (consecutive tiles stay on one XCD's L2).
    // The first D chunks of the NEXT tile are issued before the epilogue of the current one: the load latency that a
    // fresh block would sit through hides behind the epilogue's stores, which matters for the short reductions
    // (1x1 layers with C = 64 / 128: 2-4 chunks per tile).
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.kred / BKE;
    const int lcol = tid & 7;     // which 16-byte vector of the chunk
    const int lrow = tid >> 3;    // 0..RP-1
    const int HoWo = p.Ho * p.Wo;
    const size_t img_elems = (size_t)p.H * p.W * p.C;
    const int slab_stride = p.slab_bn * CHUNK_BYTES;

    // per-tile loader state (rewritten by setup())
    int m0, n0;
    __amdgpu_buffer_rsrc_t a_rsrc, b_rsrc, a2_rsrc;
    int arow[AI], ahi[AI], awi[AI];      // byte offset of the (r=0,s=0,c=0) tap; top-left input coordinate
    unsigned arow1[DUAL ? AI : 1], arow2[DUAL ? AI : 1];     // DUAL: byte offset of this row's pixel in source 1 / 2 (kOob: no row)
    const int nk1 = DUAL ? p.C / BKE : 0;                    // DUAL: chunks that come from source 1
    int kc, kr, ks, tapoff;
    // Number of the chunk being fetched.  Derived from kernel arguments only, so the weight-slab
    // offset below is provably wave-uniform (an SGPR soffset; a lane-tainted value would put every
    // buffer load into a waterfall loop).  Chunks >= nk are fetched too, but out of range: the
    // loads stay UNCONDITIONAL, the hardware returns zeros and nobody reads them.
    int uq;
    auto setup = [&](int vb) {
        const int xcd = vb & 7;
        const int q8 = ntiles >> 3, rem = ntiles & 7;
        const int tile = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (vb >> 3);
        const int tm = tile / p.tiles_n;
        const int tn = tile - tm * p.tiles_n;
        m0 = tm * BM;
        n0 = tn * BN;
        // The A descriptor is rebased at the first image of this tile so lane offsets stay < 2 GiB for
        // any batch size (a 128-row tile never spans 2 GiB of input).
        const int n_first = m0 / HoWo;
        const size_t rem_bytes = ((size_t)(p.N - n_first) * img_elems) * ES;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)p.x + (size_t)n_first * img_elems * ES), 0,
            (int)(rem_bytes > kOob ? kOob : (unsigned)rem_bytes), 0x00020000);
        // weight slabs are [n_slab][chunk][slab_bn][128 B]; a BN < slab_bn tile reads its rows inside each slab
        if constexpr (DUAL) {
            const size_t img2 = (size_t)p.H2 * p.W2 * p.C2;
            const size_t rem2 = ((size_t)(p.N - n_first) * img2) * ES;
            a2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x2 + (size_t)n_first * img2 * ES), 0,
                                                        (int)(rem2 > kOob ? kOob : (unsigned)rem2), 0x00020000);
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int m = m0 + lrow + RP * i;
                const int rl = m - n_first * HoWo;
                int nl, ho, wo;
                split_row(p, HoWo, rl, nl, ho, wo);
                const bool ok = m < p.M;
                arow1[i] = ok ? (unsigned)(rl * p.C * ES + lcol * 16) : kOob;      // source 1 is the output grid itself
                arow2[i] = ok ? (unsigned)((((nl * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.C2) * ES + lcol * 16) : kOob;
            }
        }
        const int n_in_slab = n0 % p.slab_bn;
        b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)p.w + (size_t)(n0 - n_in_slab) * p.kred * ES + (size_t)n_in_slab * CHUNK_BYTES), 0,
            (int)((unsigned)nk * (unsigned)slab_stride), 0x00020000);
        const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;       // input pixel = output pixel: no split at all
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = m0 + lrow + RP * i;
            const int rl = m - n_first * HoWo;
            if (DUAL || m >= p.M) {
                ahi[i] = -(1 << 28);
                awi[i] = 0;
                arow[i] = 0;
            } else if (pointwise) {
                ahi[i] = 0;
                awi[i] = 0;
                arow[i] = rl * p.C * ES;
            } else {
                int nl, ho, wo;
                split_row(p, HoWo, rl, nl, ho, wo);
                ahi[i] = ho * p.stride - p.pad;
                awi[i] = wo * p.stride - p.pad;
                arow[i] = (((nl * p.H + ahi[i]) * p.W + awi[i]) * p.C) * ES;
            }
        }
        // (r, s, c) of this thread's 16-byte vector inside the chunk being fetched, and its byte offset.
        // C >= one chunk: chunks are walked (r, c-chunk, s) -- exactly the order of the packed slabs.
        const int kk = lcol * EPV;
        const int pos = kk / p.C;          // C >= BKE: pos = 0
        kc = kk - pos * p.C;
        kr = pos / p.S;
        ks = pos - kr * p.S;
        tapoff = ((kr * p.W + ks) * p.C + kc) * ES;
        uq = 0;
    };
    int brow[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) brow[i] = (lrow + RP * i) * CHUNK_BYTES + lcol * 16;     // inside a [BN][128 B] slab

    // A ring of D register sets: the loads issued during chunk t are for chunk t+D, and are written to LDS
    // during chunk t+D-1 -- D-1 chunks of MFMAs of slack.  fp32 chunks are long (64 MFMAs x 64 cycles per
    // wave): D = 2 covers HBM latency; an fp16 chunk is only 16 MFMAs x 32 cycles, so it runs D = 3.
    constexpr int D = F16 ? 3 : 2;
    constexpr int P = F16 ? 6 : 2;               // unroll period = lcm(2 LDS buffers, D sets): all indices static
    f32x4 areg[D][AI], breg[D][BI];
    auto load_a = [&](f32x4 (&ar)[AI], int i) {
        if constexpr (DUAL) {      // chunk uq of source 1, or chunk uq - nk1 of source 2 (uq is wave-uniform: scalar selects)
            const bool first = uq < nk1;
            const unsigned off = (first ? arow1[i] : arow2[i]) + (unsigned)((first ? uq : uq - nk1) * CHUNK_BYTES);
            ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(first ? a_rsrc : a2_rsrc, off, 0, 0));   // ONE unconditional load
            return;
        }
        const bool ok = (unsigned)(ahi[i] + kr) < (unsigned)p.H && (unsigned)(awi[i] + ks) < (unsigned)p.W && kr < p.R;
        const unsigned off = ok ? (unsigned)(arow[i] + tapoff) : kOob;
        ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0));
    };
    auto load_b = [&](f32x4 (&br)[BI], int i) {
        const int so = uq < nk ? uq * slab_stride : (int)kOob;
        br[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, brow[i], so, 0));
    };
    auto advance_k = [&]() {     // move on by one chunk
        ++uq;
        if constexpr (DUAL) return;
        if (p.C >= BKE) {           // chunk order (r, c-chunk, s)
            if (++ks == p.S) {
                ks = 0;
                kc += BKE;
                if (kc >= p.C) { kc -= p.C; ++kr; }
            }
        } else {                    // small C (stem): (r, s, c) order, several taps per chunk
            kc += BKE;
            while (kc >= p.C) {
                kc -= p.C;
                if (++ks == p.S) { ks = 0; ++kr; }
            }
        }
        tapoff = ((kr * p.W + ks) * p.C + kc) * ES;
    };
    auto load_chunk = [&](f32x4 (&ar)[AI], f32x4 (&br)[BI]) {
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(ar, i);
#pragma unroll
        for (int i = 0; i < BI; ++i) load_b(br, i);
        advance_k();
    };
    auto store_row = [&](const f32x4 (&ar)[AI], const f32x4 (&br)[BI], int buf, int r) {
        if (r < AI) *reinterpret_cast<f32x4*>(&As[buf][(lrow + RP * r) * LDB + lcol * 16]) = ar[r];
        else *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + RP * (r - AI)) * LDB + lcol * 16]) = br[r - AI];
    };

    f32x16 acc[MT][NT];

    const int aoff = (wm0 + (lane & 31)) * LDB + (lane >> 5) * 16;
    const int boff = (wn0 + (lane & 31)) * LDB + (lane >> 5) * 16;

    // Fragment sets are double buffered ACROSS the chunk barrier: while the MFMAs of k-slot j run,
    // the ds_reads of k-slot j+1 are in flight; the next chunk is written to the other LDS buffer
    // during k-slot 2, the barrier sits before k-slot 3, and the first fragments of the next chunk
    // are fetched right behind it -- a wave never waits on LDS with an idle MFMA pipe.
    f32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    auto read_frags = [&](f32x4 (&fa)[MT], f32x4 (&fb)[NT], int buf, int k8) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&As[buf][aoff + i * 32 * LDB + k8 * 32]);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][boff + j * 32 * LDB + k8 * 32]);
    };
    auto mf = [&](const f32x4 (&fa)[MT], const f32x4 (&fb)[NT], int idx) {
        const int kk = idx / (MT * NT), i = (idx / NT) % MT, j = idx % NT;
        if constexpr (F16) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[j]),
                                                               acc[i][j], 0, 0, 0);
        } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
    };

    // One chunk of the K loop.  `ld*`: register set receiving chunk t+2; `st*`: set holding chunk t+1.
    // Side operations are spread behind individual MFMAs and never conditional.
    auto chunk = [&](int buf, f32x4 (&lda)[AI], f32x4 (&ldb)[BI], const f32x4 (&sta)[AI], const f32x4 (&stb)[BI]) {
        // k-slot 0 (+ the A gathers of chunk t+2)
        read_frags(fa1, fb1, buf, 1);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            mf(fa0, fb0, q);
#pragma unroll
            for (int i = 0; i < AI; ++i)
                if ((i * Q) / AI == q) load_a(lda, i);
        }
        // k-slot 1 (+ the weight rows of chunk t+2)
        read_frags(fa0, fb0, buf, 2);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            mf(fa1, fb1, q);
#pragma unroll
            for (int i = 0; i < BI; ++i)
                if ((i * Q) / BI == q) load_b(ldb, i);
            if (q == Q - 1) advance_k();
        }
        // k-slot 2 (+ chunk t+1 goes to the other LDS buffer)
        read_frags(fa1, fb1, buf, 3);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            mf(fa0, fb0, q);
#pragma unroll
            for (int r = 0; r < AI + BI; ++r)
                if ((r * Q) / (AI + BI) == q) store_row(sta, stb, buf ^ 1, r);
        }
        __syncthreads();
        read_frags(fa0, fb0, buf ^ 1, 0);
        // k-slot 3
#pragma unroll
        for (int q = 0; q < Q; ++q) mf(fa1, fb1, q);
    };

    setup(blockIdx.x);
#pragma unroll
    for (int d = 0; d < D; ++d) load_chunk(areg[d], breg[d]);      // chunks 0..D-1 of the first tile

    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
#pragma unroll
    for (int r = 0; r < AI + BI; ++r) store_row(areg[0], breg[0], 0, r);       // chunk 0 -> LDS buffer 0; 1..D-1 wait in registers
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();
    read_frags(fa0, fb0, 0, 0);

    // chunk t: LDS buffer t&1; loads of chunk t+D go to set t%D (its previous content, chunk t, is in LDS);
    // set (t+1)%D (chunk t+1) is written to the other LDS buffer
    for (int t = 0; t < nk; t += P) {
#pragma unroll
        for (int u = 0; u < P; ++u)
            if (t + u < nk) chunk(u & 1, areg[u % D], breg[u % D], areg[(u + 1) % D], breg[(u + 1) % D]);
    }
    __syncthreads();                 // every wave is done with both LDS buffers: the next tile's chunk 0 may land

    // ---- next tile: addressing + its first D chunks in flight before this tile's epilogue ------
    // (fp16 runs D = 3 register sets: holding them across the epilogue would spill, so it fetches after the epilogue)
    constexpr bool PREFETCH = false;    // measured: no gain on any layer shape (A/B on one box), and the D register sets held across the epilogue spill
    const int cm0 = m0, cn0 = n0;
    auto next_tile = [&]() {
        const int nvb = vb + (int)gridDim.x;
        setup(nvb < ntiles ? nvb : vb);          // last tile of this block: re-fetch the current one (never used)
#pragma unroll
        for (int d = 0; d < D; ++d) load_chunk(areg[d], breg[d]);
    };
    if constexpr (PREFETCH) next_tile();

    // ---- epilogue: y = act(acc*scale + shift + residual) --------------------------------------
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // Branch-free: residual loads / stores are raw buffer ops rebased at this tile's first row;
    // rows >= M or cols >= K get an out-of-range offset (loads return 0, stores are dropped), so all
    // residual loads of a wave are in flight together instead of one vmcnt(0) per element.
    // (the lane id goes through an opaque asm so that the per-lane offsets below are recomputed per tile instead of being
    //  hoisted out of the persistent loop, where 64 of them would stay live across the K loop)
    int lq = lane;
    asm volatile("" : "+v"(lq));
    // The epilogue at raised wave priority: beside a co-resident block that is streaming MFMAs, its ~300 instructions otherwise
    // issue one per MFMA of the partner (measured: 19-24 k cycles per epilogue, profiles/r03_igemm_swap_and_tile_timeline.txt)
    if (p.epi_prio) __builtin_amdgcn_s_setprio(3);
    const int YS = (F16 && !p.y_f32) ? 2 : 4;       // output element size
    const size_t tile_off = (size_t)cm0 * p.K;
    const unsigned rows_here = (unsigned)min(BM, p.M - cm0);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)p.y + tile_off * YS), 0, (int)(rows_here * (unsigned)p.K * YS), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(p.res ? p.res : p.y) + tile_off * ES), 0, (int)(rows_here * (unsigned)p.K * ES), 0x00020000);
    // coarse residual map (rH > 0), rebased at the first image of this tile (a tile spans at most a few images)
    const int up_first = cm0 / HoWo;
    const size_t up_img = (size_t)p.rH * p.rW * p.K * 4;
    const size_t up_rem = (size_t)(p.N - up_first) * up_img;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(p.res ? p.res : p.y) + (p.rH > 0 ? (size_t)up_first * up_img : 0)), 0,
        (int)(up_rem > kOob ? kOob : (unsigned)up_rem), 0x00020000);
    bool vec_done = false;
    if constexpr (!F16) {
        // K % 4 == 0 (every layer but the 15- / 14-wide logit heads): the accumulators go through a wave-private LDS
        // transpose so that a lane owns 4 consecutive output channels: 16-byte residual loads and stores, 4x fewer
        // vector-memory instructions than the element-wise form (the short reductions -- 1x1 layers with C = 64 / 128 --
        // spend as long in the texture path on 64 scalar stores per wave as in their MFMAs) and 256-byte row segments.
        if ((p.K & 3) == 0) {
            vec_done = true;
            constexpr int EW = WN + 4;                 // padded row, floats
            constexpr int LPR = WN / 4;                // lanes per row
            constexpr int RPI = 64 / LPR;              // rows per 16-byte pass of the wave
            constexpr int NP = 32 / RPI;               // passes per 32-row MFMA tile
            float* eb = reinterpret_cast<float*>(BM >= BN ? &As[0][0] : &Bs[0][0]) + wid * 32 * EW;
            const int er = lq / LPR, ec = (lq % LPR) * 4;
            const int n = cn0 + wn0 + ec;
            const bool nok = n < p.K;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (p.scale && nok) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
            if (p.shift && nok) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            // plain residual (the bottleneck c3 layers): a slab's residual vectors are requested BEFORE its accumulators go through
            // the LDS transpose, so their latency runs under the 64 LDS writes instead of being waited for right behind the request
            // (requesting the whole wave tile's vectors up front spills: the kernel sits at 247 VGPRs)
            const bool res_plain = p.res && p.rH == 0;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                unsigned eo[NP];
                f32x4 rv[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int row = wm0 + i * 32 + k * RPI + er;
                    eo[k] = nok ? (unsigned)(row * p.K + n) * 4u : kOob;
                }
                if (res_plain) {
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                        rv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, eo[k], 0, 0));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        eb[((r & 3) + 8 * (r >> 2) + 4 * (lq >> 5)) * EW + j * 32 + (lq & 31)] = acc[i][j][r];
                if (p.res && p.rH > 0) {
                    // nearest-upsampled residual: output pixel (img, ho, wo) reads the coarse pixel (img, ht, wt)
                    const float shs = (float)p.rH / (float)p.Ho, sws = (float)p.rW / (float)p.Wo;
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const int m = cm0 + wm0 + i * 32 + k * RPI + er;
                        int nl, ho, wo;
                        split_row(p, HoWo, m - up_first * HoWo, nl, ho, wo);
                        const int ht = min((int)floorf((float)ho * shs), p.rH - 1);
                        const int wt = min((int)floorf((float)wo * sws), p.rW - 1);
                        const unsigned uo = (nok && m < p.M) ? (unsigned)(((nl * p.rH + ht) * p.rW + wt) * p.K + n) * 4u : kOob;
                        rv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, uo, 0, 0));
                    }
                }
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&eb[(k * RPI + er) * EW + ec]) * sc + sh;
                    if (p.res) {
                        if (p.relu == 2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = rv[k][e] > 0.f ? v[e] : 0.f;    // ReLU mask of `res` (backward)
                        } else {
                            v += rv[k];
                        }
                    }
                    if (p.relu == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc, eo[k], 0, 0);
                }
            }
            __syncthreads();          // the transpose regions overlap the LDS buffers the next tile's chunk 0 goes to
        }
    }
    if constexpr (F16) {
        // fp16 output, K % 8 == 0: the same wave-private LDS transpose, a lane owns 8 consecutive output channels -- one 16-byte
        // residual load and one 16-byte store per 8 outputs instead of eight 2-byte ones (the element-wise form issues 64 scalar
        // stores per wave and MFMA tile: the 1x1 expansions with K = 256 ... 2048 spent longer in them than in their MFMAs)
        if (p.vec_epi && !p.y_f32 && (p.K & 7) == 0 && p.rH == 0) {
            vec_done = true;
            constexpr int EW = WN + 4;                 // padded row, floats
            constexpr int LPR = WN / 8;                // lanes per row
            constexpr int RPI = 64 / LPR;              // rows per 16-byte pass of the wave
            constexpr int NP = 32 / RPI;               // passes per 32-row MFMA tile
            float* eb = reinterpret_cast<float*>(BM >= BN ? &As[0][0] : &Bs[0][0]) + wid * 32 * EW;
            const int er = lq / LPR, ec = (lq % LPR) * 8;
            const int n = cn0 + wn0 + ec;
            const bool nok = n < p.K;
            f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
            if (p.scale && nok) { sc0 = *reinterpret_cast<const f32x4*>(p.scale + n); sc1 = *reinterpret_cast<const f32x4*>(p.scale + n + 4); }
            if (p.shift && nok) { sh0 = *reinterpret_cast<const f32x4*>(p.shift + n); sh1 = *reinterpret_cast<const f32x4*>(p.shift + n + 4); }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        eb[((r & 3) + 8 * (r >> 2) + 4 * (lq >> 5)) * EW + j * 32 + (lq & 31)] = acc[i][j][r];
                unsigned eo[NP];
                f32x4 rv[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int row = wm0 + i * 32 + k * RPI + er;
                    eo[k] = nok ? (unsigned)(row * p.K + n) * 2u : kOob;
                }
                if (p.res) {
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                        rv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, eo[k], 0, 0));
                }
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const float* src = &eb[(k * RPI + er) * EW + ec];
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(src) * sc0 + sh0;
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4) * sc1 + sh1;
                    if (p.res) {
                        const f16x8 rh = __builtin_bit_cast(f16x8, rv[k]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float r0 = (float)rh[e], r1 = (float)rh[e + 4];
                            v0[e] = p.relu == 2 ? (r0 > 0.f ? v0[e] : 0.f) : v0[e] + r0;
                            v1[e] = p.relu == 2 ? (r1 > 0.f ? v1[e] : 0.f) : v1[e] + r1;
                        }
                    }
                    if (p.relu == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                    }
                    f16x8 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hv[e] = (_Float16)v0[e]; hv[e + 4] = (_Float16)v1[e]; }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), y_rsrc, eo[k], 0, 0);
                }
            }
            __syncthreads();          // the transpose regions overlap the LDS buffers the next tile's chunk 0 goes to
        }
    }
    if (!vec_done)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = cn0 + wn0 + j * 32 + (lq & 31);
        const bool nok = n < p.K;
        const float sc = (p.scale && nok) ? p.scale[n] : 1.f;
        const float sh = (p.shift && nok) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            unsigned eo[16];      // element offset inside the tile
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lq >> 5);
                eo[r] = (unsigned)(row * p.K + n);
            }
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (F16) {
                        const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(r_rsrc, nok ? eo[r] * 2u : kOob, 0, 0);
                        rv[r] = (float)__builtin_bit_cast(_Float16, h);
                    } else {
                        rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, nok ? eo[r] * 4u : kOob, 0, 0));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] * sc + sh;
                if (p.res) v = p.relu == 2 ? (rv[r] > 0.f ? v : 0.f) : v + rv[r];    // relu 2: ReLU mask of `res` (backward)
                if (p.relu == 1) v = fmaxf(v, 0.f);
                if constexpr (F16) {
                    if (p.y_f32) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, nok ? eo[r] * 4u : kOob, 0, 0);
                    } else {
                        const _Float16 hv = (_Float16)v;
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), y_rsrc,
                                                              nok ? eo[r] * 2u : kOob, 0, 0);
                    }
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, nok ? eo[r] * 4u : kOob, 0, 0);
                }
            }
        }
    }
    if (p.epi_prio) __builtin_amdgcn_s_setprio(0);
    if constexpr (!PREFETCH) next_tile();
    }   // persistent tile loop
}

// =====================================================================================================================
// Split-bf16 variant ("bx3"): fp32 activations in HBM, 3 bf16 MFMAs per product, fp32 accumulate.
//   a = a_hi + a_lo (both bf16, a_hi = rn(a), a_lo = rn(a - a_hi)):  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, relative
//   error <= ~3 * 2^-18 per product (the dropped lo*lo term and the two roundings of the lo parts) -- 1e-5, two orders
//   inside the 1e-3 contract, with the fp32 exponent range (no overflow / underflow hazards of an fp16 split).
//   v_mfma_f32_32x32x16_bf16 issues 16x the FLOPs of the fp32 MFMA per cycle, so three of them per k-step are still
//   5.3x faster than the exact-fp32 path on the matrix pipe.
// Operands: the A gather is the fp32 kernel's (same buffer loads, same chunk walk: 32 k-values = 128 B per row);
//   the split of A happens ONCE per block at the LDS store (10 VALU per 16 B).  Weights are split at pack time
//   (seam_pack_conv_weight_bx3): a packed row of a chunk is [32 hi | 32 lo] bf16 = the same 128 B as an fp32 row.
// LDS: hi and lo planes, rows of exactly 64 B (32 bf16), XOR-swizzled 16-B slots (slot ^ ((row >> 2) & 3)): the
//   ds_read_b128 lane groups {0-3,12-15,20-27}/{4-11,16-19,28-31} land on 4 distinct slots => conflict-free without
//   padding; 64 KiB per block at 128x128 (two blocks per CU).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_bf16(const f32x4 v, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 x = {v[2 * p], v[2 * p + 1]};
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        const f32x2 hf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
        hi[p] = hb;
        lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, bf16x2));
    }
}

#ifndef SEAM_BX3_ABL
#define SEAM_BX3_ABL 0      // kernel experiments: 1 no global loads, 2 no LDS stores, 4 no split VALU, 8 no barrier, 16 no frag reads
#endif
template <int BM, int BN, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_igemm_bx3(const ConvArgs p) {
    constexpr int ES = 4, EPV = 4, BKE = 32;
    constexpr int WM = BM / (NW / 2), WN = BN / 2;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int RP = 8 * NW;
    constexpr int AI = BM / RP, BI = BN / RP;
    constexpr int Q = 3 * MT * NT;               // MFMAs per k-step (16 k-values)
    constexpr int PA = BM * 64 + 64;             // bytes of one A plane (+64: the lo plane starts on the other bank half)
    constexpr int PB = BN * 64 + 64;

    __shared__ __attribute__((aligned(16))) char As[2][2 * PA];     // [buffer][hi plane | lo plane]
    __shared__ __attribute__((aligned(16))) char Bs[2][2 * PB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm0 = (wid >> 1) * WM;
    const int wn0 = (wid & 1) * WN;

    const int nblk = gridDim.x;
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int q8 = nblk >> 3, rem = nblk & 7;
    const int tile = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (b >> 3);
    const int tm = tile / p.tiles_n;
    const int tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    const int nk = p.kred / BKE;
    const int lcol = tid & 7;
    const int lrow = tid >> 3;
    const int HoWo = p.Ho * p.Wo;
    const int n_first = m0 / HoWo;
    const size_t img_elems = (size_t)p.H * p.W * p.C;
    const size_t rem_bytes = ((size_t)(p.N - n_first) * img_elems) * ES;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x + (size_t)n_first * img_elems * ES), 0,
        (int)(rem_bytes > kOob ? kOob : (unsigned)rem_bytes), 0x00020000);
    const int slab_stride = p.slab_bn * CHUNK_BYTES;
    const int n_in_slab = n0 % p.slab_bn;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.w + (size_t)(n0 - n_in_slab) * p.kred * ES + (size_t)n_in_slab * CHUNK_BYTES), 0,
        (int)((unsigned)nk * (unsigned)slab_stride), 0x00020000);

    int arow[AI], ahi[AI], awi[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + lrow + RP * i;
        if (m < p.M) {
            const int n = m / HoWo;
            const int rm = m - n * HoWo;
            const int ho = rm / p.Wo;
            const int wo = rm - ho * p.Wo;
            ahi[i] = ho * p.stride - p.pad;
            awi[i] = wo * p.stride - p.pad;
            arow[i] = ((((n - n_first) * p.H + ahi[i]) * p.W + awi[i]) * p.C) * ES;
        } else {
            ahi[i] = -(1 << 28);
            awi[i] = 0;
            arow[i] = 0;
        }
    }
    int brow[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) brow[i] = (lrow + RP * i) * CHUNK_BYTES + lcol * 16;

    int kc, kr, ks, tapoff;
    {
        const int kk = lcol * EPV;
        const int pos = kk / p.C;
        kc = kk - pos * p.C;
        kr = pos / p.S;
        ks = pos - kr * p.S;
        tapoff = ((kr * p.W + ks) * p.C + kc) * ES;
    }
    int uq = 0;

    constexpr int D = 2;
    f32x4 areg[D][AI], breg[D][BI];
    auto load_a = [&](f32x4 (&ar)[AI], int i) {
        const bool ok = (unsigned)(ahi[i] + kr) < (unsigned)p.H && (unsigned)(awi[i] + ks) < (unsigned)p.W && kr < p.R;
        const unsigned off = ok ? (unsigned)(arow[i] + tapoff) : kOob;
        if (!(SEAM_BX3_ABL & 1)) ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0));
    };
    auto load_b = [&](f32x4 (&br)[BI], int i) {
        const int so = uq < nk ? uq * slab_stride : (int)kOob;
        if (!(SEAM_BX3_ABL & 1)) br[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, brow[i], so, 0));
    };
    auto advance_k = [&]() {
        ++uq;
        if (p.C >= BKE) {
            if (++ks == p.S) {
                ks = 0;
                kc += BKE;
                if (kc >= p.C) { kc -= p.C; ++kr; }
            }
        } else {
            kc += BKE;
            while (kc >= p.C) {
                kc -= p.C;
                if (++ks == p.S) { ks = 0; ++kr; }
            }
        }
        tapoff = ((kr * p.W + ks) * p.C + kc) * ES;
    };
    auto load_chunk = [&](f32x4 (&ar)[AI], f32x4 (&br)[BI]) {
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(ar, i);
#pragma unroll
        for (int i = 0; i < BI; ++i) load_b(br, i);
        advance_k();
    };
    // LDS addresses of this thread's stores.  A: 4 floats -> 8 B of hi + 8 B of lo at bf16 index 4*lcol of the row;
    // B: 16 B of the packed row: lcol < 4 -> hi plane slot lcol, else lo plane slot lcol-4.
    auto store_row = [&](const f32x4 (&ar)[AI], const f32x4 (&br)[BI], int buf, int r) {
        if (SEAM_BX3_ABL & 2) return;
        if (r < AI) {
            const int row = lrow + RP * r;
            const int off = row * 64 + ((((lcol >> 1) ^ (row >> 2)) & 3) << 4) + (lcol & 1) * 8;
            u32x2 hi, lo;
            if (SEAM_BX3_ABL & 4) { hi[0] = __builtin_bit_cast(unsigned, ar[r][0]); hi[1] = __builtin_bit_cast(unsigned, ar[r][1]);
                                    lo[0] = __builtin_bit_cast(unsigned, ar[r][2]); lo[1] = __builtin_bit_cast(unsigned, ar[r][3]); }
            else split_bf16(ar[r], hi, lo);
            *reinterpret_cast<u32x2*>(&As[buf][off]) = hi;
            *reinterpret_cast<u32x2*>(&As[buf][PA + off]) = lo;
        } else {
            const int row = lrow + RP * (r - AI);
            const int off = (lcol >> 2) * PB + row * 64 + (((lcol ^ (row >> 2)) & 3) << 4);
            *reinterpret_cast<f32x4*>(&Bs[buf][off]) = br[r - AI];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: lane reads 16 B (8 bf16 of k) of row (l & 31) at slot 2*step + (l >> 5), swizzled
    const int arow0 = wm0 + (lane & 31), brow0 = wn0 + (lane & 31);
    struct Frag { f32x4 ah[MT], al[MT], bh[NT], bl[NT]; };
    Frag f0, f1;
    auto read_frags = [&](Frag& f, int buf, int step) {
        if (SEAM_BX3_ABL & 16) return;
        const int slot = 2 * step + (lane >> 5);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = arow0 + i * 32;
            const int off = row * 64 + (((slot ^ (row >> 2)) & 3) << 4);
            f.ah[i] = *reinterpret_cast<const f32x4*>(&As[buf][off]);
            f.al[i] = *reinterpret_cast<const f32x4*>(&As[buf][PA + off]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int row = brow0 + j * 32;
            const int off = row * 64 + (((slot ^ (row >> 2)) & 3) << 4);
            f.bh[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][off]);
            f.bl[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][PB + off]);
        }
    };
    auto mf = [&](const Frag& f, int idx) {       // idx in [0, Q): term-major so the small terms of a tile go in first
        const int term = idx / (MT * NT), i = (idx / NT) % MT, j = idx % NT;
        const f32x4 a = term == 1 ? f.al[i] : f.ah[i];
        const f32x4 bq = term == 0 ? f.bl[j] : f.bh[j];          // 0: hi*lo, 1: lo*hi, 2: hi*hi
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq),
                                                            acc[i][j], 0, 0, 0);
    };

    auto chunk = [&](int buf, f32x4 (&lda)[AI], f32x4 (&ldb)[BI], const f32x4 (&sta)[AI], const f32x4 (&stb)[BI]) {
        // k-step 0 (+ the gathers and weight rows of chunk t+D)
        read_frags(f1, buf, 1);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            mf(f0, q);
#pragma unroll
            for (int i = 0; i < AI + BI; ++i)
                if ((i * Q) / (AI + BI) == q) {
                    if (i < AI) load_a(lda, i);
                    else load_b(ldb, i - AI);
                }
            if (q == Q - 1) advance_k();
        }
        // k-step 1: chunk t+1 goes to the other LDS buffer behind the first two thirds of the MFMAs, then the barrier
        // and the first fragments of the next chunk; the last third covers their latency
        constexpr int QS = (2 * Q) / 3;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            mf(f1, q);
#pragma unroll
            for (int r = 0; r < AI + BI; ++r)
                if ((r * QS) / (AI + BI) == q) store_row(sta, stb, buf ^ 1, r);
            if (q == QS - 1) {
                if (!(SEAM_BX3_ABL & 8)) __syncthreads();
                read_frags(f0, buf ^ 1, 0);
            }
        }
    };

    load_chunk(areg[0], breg[0]);
#pragma unroll
    for (int r = 0; r < AI + BI; ++r) store_row(areg[0], breg[0], 0, r);
    load_chunk(areg[1], breg[1]);
    __syncthreads();
    read_frags(f0, 0, 0);

    for (int t = 0; t < nk; t += 2) {
        chunk(0, areg[0], breg[0], areg[1], breg[1]);
        if (t + 1 < nk) chunk(1, areg[1], breg[1], areg[0], breg[0]);
    }

    // ---- epilogue (the fp32 kernel's) ---------------------------------------------------------
    const size_t tile_off = (size_t)m0 * p.K;
    const unsigned rows_here = (unsigned)min(BM, p.M - m0);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)p.y + tile_off * 4), 0, (int)(rows_here * (unsigned)p.K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(p.res ? p.res : p.y) + tile_off * 4), 0, (int)(rows_here * (unsigned)p.K * 4), 0x00020000);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const bool nok = n < p.K;
        const float sc = (p.scale && nok) ? p.scale[n] : 1.f;
        const float sh = (p.shift && nok) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            unsigned eo[16];
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                eo[r] = (unsigned)(row * p.K + n);
            }
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, nok ? eo[r] * 4u : kOob, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] * sc + sh;
                if (p.res) v = p.relu == 2 ? (rv[r] > 0.f ? v : 0.f) : v + rv[r];
                if (p.relu == 1) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, nok ? eo[r] * 4u : kOob, 0, 0);
            }
        }
    }
}

// Re-layout of fp32-packed weights [..][row][32 floats] -> [..][row][32 hi bf16 | 32 lo bf16] (same 128 B per row-chunk)
__global__ void split_weight_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float x = in[i];
        const __bf16 h = (__bf16)x;
        const __bf16 l = (__bf16)(x - (float)h);
        const size_t rowchunk = i >> 5;
        const int col = (int)(i & 31);
        out[rowchunk * 64 + col] = __builtin_bit_cast(unsigned short, h);
        out[rowchunk * 64 + 32 + col] = __builtin_bit_cast(unsigned short, l);
    }
}

// OIHW fp32 [K,Cin,R,S] -> tile-contiguous packed weights of type T: [rows/BN][kred/BKE][BN][BKE].
// Chunk q of the reduction covers, for Cstore >= BKE: tap row r = q / (S*Cstore/BKE), channel chunk
// cc = (q / S) % (Cstore/BKE), tap column s = q % S, channels cc*BKE ..; for Cstore < BKE (stem):
// reduction index kk = q*BKE + col in plain (r, s, c) order.
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int K, int Cin, int R, int S,
                                   int Cstore, int kred, int rows, int bn, int mode) {
    constexpr int BKE = CHUNK_BYTES / (int)sizeof(T);
    const size_t total = (size_t)rows * kred;
    const int nk = kred / BKE;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % BKE);
        size_t rest = i / BKE;
        const int row_in = (int)(rest % bn);
        rest /= bn;
        const int q = (int)(rest % nk);
        const int tn = (int)(rest / nk);
        const int row = tn * bn + row_in;
        int r, s2, c;
        if (Cstore >= BKE) {
            const int ccn = Cstore / BKE;
            s2 = q % S;
            c = ((q / S) % ccn) * BKE + col;
            r = q / (S * ccn);
        } else {
            const int kk = q * BKE + col;
            const int pos = kk / Cstore;
            c = kk - pos * Cstore;
            r = pos / S;
            s2 = pos - r * S;
        }
        float v = 0.f;
        if (row < K && c < Cin && r < R) {
            if (mode == 0) {
                v = w[(((size_t)row * Cin + c) * R + r) * S + s2];
            } else if (mode == 2) {   // input-gradient weights of a Conv2d [Cin(out), K(in), R, S]: rotate 180, swap channels
                v = w[(((size_t)c * K + row) * R + (R - 1 - r)) * S + (S - 1 - s2)];
            } else {  // ConvTranspose2d weight [Cin, Cout, 2, 2]; row = (a*2+b)*Cout + co
                const int cout = K / 4;
                const int ab = row / cout;
                const int co = row - ab * cout;
                v = w[(((size_t)c * cout + co) * 2 + (ab >> 1)) * 2 + (ab & 1)];
            }
        }
        out[i] = (T)v;
    }
}

template <typename T>
int kred_of(int C, int R, int S) {
    constexpr int BKE = CHUNK_BYTES / (int)sizeof(T);
    return ((R * S * C + BKE - 1) / BKE) * BKE;
}

template <typename T>
int pack_weight(const float* w, void* w_packed, int K, int Cin, int R, int S, int Cstore, int mode, void* stream) {
    constexpr int BKE = CHUNK_BYTES / (int)sizeof(T), EPV = 16 / (int)sizeof(T);
    if ((Cstore % EPV) || (Cstore >= BKE && Cstore % BKE)) return (int)hipErrorInvalidValue;
    const int kred = kred_of<T>(Cstore, R, S);
    const int rows = ((K + 63) / 64) * 64;
    const size_t total = (size_t)rows * kred;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (T*)w_packed, K, Cin, R, S,
                       Cstore, kred, rows, rows % 128 == 0 ? 128 : 64, mode);
    return (int)hipGetLastError();
}

// Tile choice.  Cost model: ceil(blocks / 256 CUs) * BM * BN, weighted by the measured relative cost per FLOP of each
// tile shape (smaller tiles move more operand bytes per FLOP).  What the probes say about the load path
// (tools/l2_bw_probe.hip, tools/pmc_conv.sh): L2 hits stream at 48-55 B/clk/CU, L2 misses at 11.6; the conv kernels
// sit at ~16 because 13-16 % of their 128-B requests miss the 4 MiB L2 (each input line is re-fetched once per tap row
// r by tiles that touch it ~40 us apart), so halving the weight traffic with the 256x128 / 8-wave tile buys only
// 2-6 % (fp16 / split-bf16) and nothing for exact fp32, which is MFMA-bound.  The lever that remains is staging the
// input patch of a tile in LDS once per channel chunk and running all taps from it (next round).
enum Prec { P_F32 = 0, P_F16 = 1, P_BX3 = 2 };

inline int tile_weight(int prec, int bm, int bn, int taps) {
    const int area = bm * bn;
    if (prec == P_F32) return area == 256 * 128 ? 100 : area == 128 * 128 ? 100 : area == 64 * 64 ? 125 : 110;
    // fp16: the 256 x 128 / 8-wave tile wins on the 3x3 layers (859 vs 752 TFLOP/s at 80 x 200^2 x 256 -> 256, +3-5 % at 100^2 and on
    // the 14 x 14 ROI maps) and loses 3-5 % on the 1x1 layers (profiles/r02_f16_wave128.txt)
    if (prec == P_F16 && area == 256 * 128) return taps >= 9 ? 92 : 105;
    return area == 256 * 128 ? 95 : area == 128 * 128 ? 100 : area == 64 * 64 ? 200 : 150;
}

inline void choose_tile(int prec, int M, int K, int& best_bm, int& best_bn, int taps = 1) {
    const int rows = ((K + 63) / 64) * 64;
    const int slab = rows % 128 == 0 ? 128 : 64;
    const int force = seam_opt::get(seam_opt::CONV_TILE);           // kernel experiments: BM * 1000 + BN (256128, 128128, ...)
    if (force) {
        const int fm = force / 1000, fn = force % 1000;
        if ((fm == 256 || fm == 128 || fm == 64) && (fn == 128 || fn == 64) &&
            fn <= slab && (fm != 256 || fn == 128)) {
            best_bm = fm; best_bn = fn;
            return;
        }
    }
    long best_cost = -1;
    best_bm = 128;
    best_bn = slab;
    for (int bm = 256; bm >= 64; bm >>= 1)
        for (int bn = slab; bn >= 64; bn -= 64) {
            if (bm == 256 && (bn != 128 || prec == P_F32)) continue;       // 8-wave tile: 256x128 only; no gain for exact fp32
            const long nb = (long)((M + bm - 1) / bm) * (rows / bn);
            const long cost = ((nb + 255) / 256) * bm * bn * tile_weight(prec, bm, bn, taps);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_bm = bm; best_bn = bn; }
        }
}

struct DualSrc { const void* x2; int H2, W2, C2, stride2; };

// multipliers of split_row(): exact while the operand times the divisor stays below 2^32 -- rl < Ho*Wo + 256 against Ho*Wo (used
// only when Ho*Wo < 256), rm < Ho*Wo against Wo
inline void set_row_split(ConvArgs& a) {
    const unsigned long long howo = (unsigned long long)a.Ho * a.Wo;
    auto magic = [](unsigned long long d) -> unsigned { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); };
    a.m_HoWo = magic(howo);
    a.m_Wo = magic((unsigned long long)a.Wo);
    a.div_fast = (a.Ho > 0 && a.Wo > 0 && howo * (unsigned long long)a.Wo < (1ull << 32)) ? 1 : 0;
}

template <typename T>
int conv2d(const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual, void* y,
           int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int relu, int y_f32, void* stream,
           int rH = 0, int rW = 0, const DualSrc* dual = nullptr, int ho_crop = 0, int wo_crop = 0) {
    constexpr int BKE = CHUNK_BYTES / (int)sizeof(T), EPV = 16 / (int)sizeof(T);
    if ((C % EPV) || (C >= BKE && C % BKE) || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
    ConvArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.H = H; a.W = W; a.C = C;
    a.Ho = (H + 2 * pad - R) / stride + 1;
    a.Wo = (W + 2 * pad - S) / stride + 1;
    a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
    if (ho_crop > 0 && wo_crop > 0) {          // top-left crop of the output grid (asymmetric padding: `pad` before, less after)
        if (ho_crop > a.Ho || wo_crop > a.Wo) return (int)hipErrorInvalidValue;
        a.Ho = ho_crop; a.Wo = wo_crop;
    }
    a.kred = kred_of<T>(C, R, S);
    a.M = N * a.Ho * a.Wo;
    a.relu = relu;
    a.y_f32 = y_f32;
    a.vec_epi = seam_opt::get(seam_opt::F16_VEC_EPILOGUE);
    a.epi_prio = seam_opt::get(seam_opt::EPI_PRIO);

    a.rH = rH; a.rW = rW;
    a.x2 = nullptr; a.H2 = a.W2 = a.C2 = a.stride2 = 0;
    set_row_split(a);
    if (dual) {
        if (R != 1 || S != 1 || stride != 1 || pad != 0 || (C % BKE) || (dual->C2 % BKE) || dual->stride2 < 1 ||
            (a.Ho - 1) * dual->stride2 >= dual->H2 || (a.Wo - 1) * dual->stride2 >= dual->W2 || !dual->x2)
            return (int)hipErrorInvalidValue;
        a.x2 = dual->x2; a.H2 = dual->H2; a.W2 = dual->W2; a.C2 = dual->C2; a.stride2 = dual->stride2;
        a.kred = C + dual->C2;
    }
    if (a.Ho <= 0 || a.Wo <= 0) return (int)hipErrorInvalidValue;
    if (rH > 0 && (sizeof(T) != 4 || (K & 3) || rW <= 0 || !residual || relu == 2)) return (int)hipErrorInvalidValue;
    const int rows = ((K + 63) / 64) * 64;
    a.slab_bn = rows % 128 == 0 ? 128 : 64;
    int best_bm, best_bn;
    choose_tile(sizeof(T) == 2 ? P_F16 : P_F32, a.M, K, best_bm, best_bn, R * S);
    a.tiles_m = (a.M + best_bm - 1) / best_bm;
    a.tiles_n = rows / best_bn;
    // persistent blocks: one per resident slot (256 CUs x 2 blocks of 4 waves, or x 1 block of 8 waves); a multiple of 8 so
    // that a block's tiles stay on its XCD
    const int slots4 = seam_opt::get(seam_opt::CONV_SLOTS);      // dev knob
    const int ntiles = a.tiles_m * a.tiles_n;
    const int slots = best_bm == 256 ? slots4 / 2 : slots4;
    const dim3 grid(ntiles < slots ? ntiles : slots);
    const int dyn = seam_opt::get(seam_opt::CONV_DYNLDS);   // dev knob: occupancy experiments
    hipStream_t st = (hipStream_t)stream;
    {
        if (dual) {
            if (best_bm == 256) {                                      // (fp16 only) the dual form has no 8-wave instance
                best_bm = 128;
                a.tiles_m = (a.M + best_bm - 1) / best_bm;
                const int nt2 = a.tiles_m * a.tiles_n;
                const dim3 grid2(nt2 < slots4 ? nt2 : slots4);
                hipLaunchKernelGGL((conv_igemm<T, 128, 128, 4, true>), grid2, dim3(256), dyn, st, a);
                return (int)hipGetLastError();
            }
            if (best_bm == 128 && best_bn == 128) hipLaunchKernelGGL((conv_igemm<T, 128, 128, 4, true>), grid, dim3(256), dyn, st, a);
            else if (best_bm == 128) hipLaunchKernelGGL((conv_igemm<T, 128, 64, 4, true>), grid, dim3(256), dyn, st, a);
            else if (best_bn == 128) hipLaunchKernelGGL((conv_igemm<T, 64, 128, 4, true>), grid, dim3(256), dyn, st, a);
            else hipLaunchKernelGGL((conv_igemm<T, 64, 64, 4, true>), grid, dim3(256), dyn, st, a);
            return (int)hipGetLastError();
        }
    }
    if (best_bm == 256) hipLaunchKernelGGL((conv_igemm<T, 256, 128, 8>), grid, dim3(512), dyn, st, a);
    else if (best_bm == 128 && best_bn == 128) hipLaunchKernelGGL((conv_igemm<T, 128, 128, 4>), grid, dim3(256), dyn, st, a);
    else if (best_bm == 128) hipLaunchKernelGGL((conv_igemm<T, 128, 64, 4>), grid, dim3(256), dyn, st, a);
    else if (best_bn == 128) hipLaunchKernelGGL((conv_igemm<T, 64, 128, 4>), grid, dim3(256), dyn, st, a);
    else hipLaunchKernelGGL((conv_igemm<T, 64, 64, 4>), grid, dim3(256), dyn, st, a);
    return (int)hipGetLastError();
}

int conv2d_bx3(const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual, void* y,
               int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int relu, void* stream) {
    if ((C % 4) || (C >= 32 && C % 32) || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
    ConvArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.H = H; a.W = W; a.C = C;
    a.Ho = (H + 2 * pad - R) / stride + 1;
    a.Wo = (W + 2 * pad - S) / stride + 1;
    a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
    a.kred = kred_of<float>(C, R, S);
    a.M = N * a.Ho * a.Wo;
    a.relu = relu;
    a.y_f32 = 1;
    a.vec_epi = 0;
    a.epi_prio = 0;
    a.rH = 0; a.rW = 0;
    a.x2 = nullptr; a.H2 = a.W2 = a.C2 = a.stride2 = 0;
    set_row_split(a);
    if (a.Ho <= 0 || a.Wo <= 0) return (int)hipErrorInvalidValue;
    const int rows = ((K + 63) / 64) * 64;
    a.slab_bn = rows % 128 == 0 ? 128 : 64;
    int best_bm, best_bn;
    choose_tile(P_BX3, a.M, K, best_bm, best_bn);
    a.tiles_m = (a.M + best_bm - 1) / best_bm;
    a.tiles_n = rows / best_bn;
    const dim3 grid(a.tiles_m * a.tiles_n);
    hipStream_t st = (hipStream_t)stream;
    if (best_bm == 256) hipLaunchKernelGGL((conv_igemm_bx3<256, 128, 8>), grid, dim3(512), 0, st, a);
    else if (best_bm == 128 && best_bn == 128) hipLaunchKernelGGL((conv_igemm_bx3<128, 128, 4>), grid, dim3(256), 0, st, a);
    else if (best_bm == 128) hipLaunchKernelGGL((conv_igemm_bx3<128, 64, 4>), grid, dim3(256), 0, st, a);
    else if (best_bn == 128) hipLaunchKernelGGL((conv_igemm_bx3<64, 128, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_bx3<64, 64, 4>), grid, dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int seam_conv_kred(int C, int R, int S) { return kred_of<float>(C, R, S); }
int seam_conv_kred_f16(int C, int R, int S) { return kred_of<_Float16>(C, R, S); }
int seam_conv_rows_padded(int K) { return ((K + 63) / 64) * 64; }
int seam_conv_tile(int M, int K) {      // BM * 1000 + BN the fp32 launcher will pick for an [M x K] output
    int bm, bn;
    choose_tile(P_F32, M, K, bm, bn);
    return bm * 1000 + bn;
}
int seam_conv_tile_prec(int prec, int M, int K) {      // same for prec 0 fp32 | 1 fp16 | 2 split-bf16 (a 1x1 layer)
    int bm, bn;
    choose_tile(prec, M, K, bm, bn);
    return bm * 1000 + bn;
}
int seam_conv_tile_taps(int prec, int M, int K, int taps) {      // ... of a layer with R*S = taps (the fp16 choice depends on it)
    int bm, bn;
    choose_tile(prec, M, K, bm, bn, taps);
    return bm * 1000 + bn;
}

int seam_pack_conv_weight_f32(const float* w, float* w_packed, int K, int Cin, int R, int S, int Cstore, int mode,
                              void* stream) {
    return pack_weight<float>(w, w_packed, K, Cin, R, S, Cstore, mode, stream);
}

int seam_pack_conv_weight_f16(const float* w, void* w_packed, int K, int Cin, int R, int S, int Cstore, int mode,
                              void* stream) {
    return pack_weight<_Float16>(w, w_packed, K, Cin, R, S, Cstore, mode, stream);
}

int seam_conv2d_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                    const float* residual, float* y, int N, int H, int W, int C, int K, int R, int S, int stride,
                    int pad, int relu, void* stream) {
    return conv2d<float>(x, w_packed, scale, shift, residual, y, N, H, W, C, K, R, S, stride, pad, relu, 1, stream);
}

int seam_conv2d_upres_f32(const float* x, const float* w_packed, const float* scale, const float* shift, const float* top,
                          float* y, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ht, int Wt,
                          int relu, void* stream) {
    if (Ht <= 0 || Wt <= 0) return (int)hipErrorInvalidValue;
    return conv2d<float>(x, w_packed, scale, shift, top, y, N, H, W, C, K, R, S, stride, pad, relu, 1, stream, Ht, Wt);
}

int seam_conv2d_crop_f32(const float* x, const float* w_packed, const float* scale, const float* shift, float* y, int N, int H,
                         int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo, int relu, void* stream) {
    if (Ho <= 0 || Wo <= 0) return (int)hipErrorInvalidValue;
    return conv2d<float>(x, w_packed, scale, shift, nullptr, y, N, H, W, C, K, R, S, stride, pad, relu, 1, stream, 0, 0, nullptr, Ho, Wo);
}

int seam_conv2d_crop_f16(const void* x, const void* w_packed, const float* scale, const float* shift, void* y, int N, int H, int W,
                         int C, int K, int R, int S, int stride, int pad, int Ho, int Wo, int relu, void* stream) {
    if (Ho <= 0 || Wo <= 0) return (int)hipErrorInvalidValue;
    return conv2d<_Float16>(x, w_packed, scale, shift, nullptr, y, N, H, W, C, K, R, S, stride, pad, relu, 0, stream, 0, 0, nullptr, Ho, Wo);
}

int seam_conv2d_dual_f16(const void* x1, const void* x2, const void* w_packed, const float* scale, const float* shift, void* y,
                         int N, int Ho, int Wo, int C1, int H2, int W2, int C2, int stride2, int K, int relu, void* stream) {
    const DualSrc d = {x2, H2, W2, C2, stride2};
    return conv2d<_Float16>(x1, w_packed, scale, shift, nullptr, y, N, Ho, Wo, C1, K, 1, 1, 1, 0, relu, 0, stream, 0, 0, &d);
}

int seam_conv2d_dual_f32(const float* x1, const float* x2, const float* w_packed, const float* scale, const float* shift,
                         float* y, int N, int Ho, int Wo, int C1, int H2, int W2, int C2, int stride2, int K, int relu,
                         void* stream) {
    const DualSrc d = {x2, H2, W2, C2, stride2};
    return conv2d<float>(x1, w_packed, scale, shift, nullptr, y, N, Ho, Wo, C1, K, 1, 1, 1, 0, relu, 1, stream, 0, 0, &d);
}

int seam_conv2d_f16(const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual,
                    void* y, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int relu,
                    int y_f32, void* stream) {
    return conv2d<_Float16>(x, w_packed, scale, shift, residual, y, N, H, W, C, K, R, S, stride, pad, relu, y_f32, stream);
}

/* Split-bf16 path: weights = seam_pack_conv_weight_f32's layout with every 128-byte row-chunk re-written as
 * [32 hi bf16 | 32 lo bf16]; `tmp` = rows_padded*kred floats of scratch (the fp32 pack). */
int seam_pack_conv_weight_bx3(const float* w, void* w_packed, float* tmp, int K, int Cin, int R, int S, int Cstore, int mode,
                              void* stream) {
    const int rc = pack_weight<float>(w, tmp, K, Cin, R, S, Cstore, mode, stream);
    if (rc) return rc;
    const size_t total = (size_t)(((K + 63) / 64) * 64) * kred_of<float>(Cstore, R, S);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(split_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tmp, (unsigned short*)w_packed, total);
    return (int)hipGetLastError();
}

int seam_conv2d_bx3(const float* x, const void* w_packed, const float* scale, const float* shift, const float* residual,
                    float* y, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int relu, void* stream) {
    return conv2d_bx3(x, w_packed, scale, shift, residual, y, N, H, W, C, K, R, S, stride, pad, relu, stream);
}

}  // extern "C"
