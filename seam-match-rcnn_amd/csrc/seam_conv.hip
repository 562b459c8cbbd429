// seam_conv.hip -- implicit-GEMM convolution on gfx950, exact fp32 on the matrix cores.
//
// One kernel family serves every dense contraction of the path (ResNet-50 body, FPN, RPN head,
// box/mask heads, the match trunk's valid 3x3 convs and its Linear) -- see include/seam_hip.h.
//
// GEMM view (TN):  Y[M, K] = A[M, kred] * B[K, kred]^T
//   M    = N*Ho*Wo output pixels, row m -> (n, ho, wo)          (NHWC output == row-major [M][K])
//   kred = reduction index, walked in 32-wide chunks ordered (r, c-chunk, s) [C >= 32] so the three
//          horizontal taps of one (row, channel-chunk) are consecutive chunks and re-hit the same
//          A lines in L1; A is gathered on the fly from the NHWC input (hardware zero fill for
//          padding / tails); B = pre-packed weights, stored TILE-CONTIGUOUS: [n_tile][chunk][BN][32]
//          (one 16 KiB slab per chunk: no power-of-two row stride, no set conflicts, 4 TLB pages).
// Tiling for CDNA4 (wave64, 4 SIMDs/CU):
//   block 256 threads = 4 waves (2x2); block tile BM x BN x 32; wave tile (BM/2) x (BN/2) built
//   from 32x32 v_mfma_f32_32x32x2_f32 tiles (16 accumulator VGPRs each, 64 cyc/issue = the fp32 rate;
//   bit-exact fp32 fma chain).  Operands go global -> registers -> LDS (rows padded to 36 floats:
//   9 x 16 B slots, odd => ds_read_b128 / ds_write_b128 lane groups are conflict-free), double
//   buffered, next chunk's global loads in flight under the current chunk's 64 MFMAs per wave.
//   A fragment: lane l reads row (l&31), k-quad (l>>5): one ds_read_b128 feeds 4 MFMAs
//   (k-pairs {j, j+4}); the same permutation is applied to B, so the contraction is unchanged.
//   blockIdx -> tile mapping is XCD-aware: consecutive tiles (same A rows, neighbouring halos)
//   stay on one XCD's L2 (dispatch is round-robin b % 8, guide T1, bijective form).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;   // padded LDS row (floats)

struct ConvArgs {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int N, H, W, C;
    int Ho, Wo, K;
    int R, S, stride, pad;
    int kred;
    int M;
    int relu;
    int tiles_m, tiles_n;
};

template <int BM, int BN, int V = 0>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(const ConvArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2;      // wave tile
    constexpr int MT = WM / 32, NT = WN / 32;    // 32x32 MFMA tiles per wave
    constexpr int AI = BM / 32, BI = BN / 32;    // float4 loads per thread per chunk

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm0 = (wid >> 1) * WM;
    const int wn0 = (wid & 1) * WN;

    // ---- XCD-aware tile id (bijective for any grid size) --------------------------------------
    const int nblk = gridDim.x;
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int q = nblk >> 3, rem = nblk & 7;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    const int tm = tile / p.tiles_n;
    const int tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    // ---- loader state -------------------------------------------------------------------------
    // Both operands are fetched with raw buffer loads (SGPR descriptor + 32-bit lane offset): no
    // 64-bit address math in the loop, and an out-of-image tap / tail row simply uses an offset
    // beyond num_records, for which the hardware returns zeros (no select, no branch).
    // The A descriptor is rebased at the first image of this tile so lane offsets stay < 2 GiB for
    // any batch size (a 128-row tile never spans 2 GiB of input).
    constexpr unsigned kOob = 0x80000000u;
    const int nk = p.kred / BK;
    const int lcol = tid & 7;     // which float4 of the 32-wide k chunk
    const int lrow = tid >> 3;    // 0..31
    const int HoWo = p.Ho * p.Wo;
    const int n_first = m0 / HoWo;
    const size_t img_elems = (size_t)p.H * p.W * p.C;
    const size_t rem_bytes = ((size_t)(p.N - n_first) * img_elems) * sizeof(float);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (size_t)n_first * img_elems), 0, (int)(rem_bytes > kOob ? kOob : (unsigned)rem_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.w + (size_t)n0 * p.kred), 0, (int)((unsigned)BN * (unsigned)p.kred * 4u), 0x00020000);   // tile tn

    int arow[AI], ahi[AI], awi[AI];      // byte offset of the (r=0,s=0,c=0) tap; top-left input coordinate
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            const int n = m / HoWo;
            const int rm = m - n * HoWo;
            const int ho = rm / p.Wo;
            const int wo = rm - ho * p.Wo;
            ahi[i] = ho * p.stride - p.pad;
            awi[i] = wo * p.stride - p.pad;
            arow[i] = ((((n - n_first) * p.H + ahi[i]) * p.W + awi[i]) * p.C) * 4;
        } else {
            ahi[i] = -(1 << 28);
            awi[i] = 0;
            arow[i] = 0;
        }
    }
    int brow[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) brow[i] = ((lrow + 32 * i) * BK + lcol * 4) * 4;     // inside a [BN][32] slab

    // (r, s, c) of this thread's float4 inside the chunk being fetched, and its byte offset.
    // C >= 32: chunks are walked (r, c-chunk, s) -- exactly the order of the packed weight slabs.
    int kc, kr, ks, tapoff;
    {
        const int kk = lcol * 4;
        const int pos = kk / p.C;          // C >= 32: pos = 0
        kc = kk - pos * p.C;
        kr = pos / p.S;
        ks = pos - kr * p.S;
        tapoff = ((kr * p.W + ks) * p.C + kc) * 4;
    }
    // Number of the chunk being fetched.  Derived from kernel arguments only, so the weight-slab
    // offset below is provably wave-uniform (an SGPR soffset; a lane-tainted value would put every
    // buffer load into a waterfall loop).  Chunks >= nk are fetched too, but out of range: the
    // loads stay UNCONDITIONAL (a branch around a load makes hipcc wait vmcnt(0) right behind it,
    // serialising the whole prefetch), the hardware returns zeros and nobody reads them.
    int uq = 0;

    // Two register sets: the loads issued during chunk t are for chunk t+2 (a whole chunk of
    // MFMAs of slack before they are written to LDS during chunk t+1).
    f32x4 areg0[AI], breg0[BI], areg1[AI], breg1[BI];
    auto load_a = [&](f32x4 (&ar)[AI], int i) {
        const bool ok = (unsigned)(ahi[i] + kr) < (unsigned)p.H && (unsigned)(awi[i] + ks) < (unsigned)p.W && kr < p.R;
        unsigned off = ok ? (unsigned)(arow[i] + tapoff) : kOob;
        if (V & 16) off = lcol * 16;            // dev ablation: A always hits the same 128 B line
        if (V & 1) off = kOob;                  // dev ablation: no A traffic
        ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0));
    };
    auto load_b = [&](f32x4 (&br)[BI], int i) {
        const int so = uq < nk ? uq * (BN * BK * 4) : (int)kOob;
        br[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (V & 32) ? lcol * 16 : brow[i],
                                                                               (V & 32) ? 0 : so, 0));
    };
    auto advance_k = [&]() {     // move on by one chunk
        ++uq;
        if (p.C >= BK) {            // chunk order (r, c-chunk, s)
            if (++ks == p.S) {
                ks = 0;
                kc += BK;
                if (kc >= p.C) { kc -= p.C; ++kr; }
            }
        } else {                    // small C (stem): (r, s, c) order, several taps per chunk
            kc += BK;
            while (kc >= p.C) {
                kc -= p.C;
                if (++ks == p.S) { ks = 0; ++kr; }
            }
        }
        tapoff = ((kr * p.W + ks) * p.C + kc) * 4;
    };
    auto load_chunk = [&](f32x4 (&ar)[AI], f32x4 (&br)[BI]) {
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(ar, i);
#pragma unroll
        for (int i = 0; i < BI; ++i) load_b(br, i);
        advance_k();
    };
    auto store_chunk = [&](const f32x4 (&ar)[AI], const f32x4 (&br)[BI], int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32 * i) * LDK + lcol * 4]) = ar[i];
#pragma unroll
        for (int i = 0; i < BI; ++i)
            *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + 32 * i) * LDK + lcol * 4]) = br[i];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31;
    const int fk = (lane >> 5) * 4;
    const int aoff = (wm0 + frow) * LDK + fk;
    const int boff = (wn0 + frow) * LDK + fk;

    // Fragment sets are double buffered ACROSS the chunk barrier: while the MFMAs of k-step j run,
    // the ds_reads of k-step j+1 are in flight; the next chunk is written to the other LDS buffer
    // in the middle of k-step 2, the barrier sits before k-step 3, and the first fragments of the
    // next chunk are fetched right behind it -- so a wave never waits on LDS with an idle MFMA pipe.
    f32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    auto read_frags = [&](f32x4 (&fa)[MT], f32x4 (&fb)[NT], int buf, int k8) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&As[buf][aoff + i * 32 * LDK + k8 * 8]);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][boff + j * 32 * LDK + k8 * 8]);
    };
    // One chunk of the K loop.  `ld*`: register set receiving chunk t+2; `st*`: set holding chunk t+1.
    // Every side operation (buffer load, LDS write) is placed behind its own MFMA: a VMEM / wide DS
    // instruction costs tens of issue cycles, and two of them back to back leave the matrix pipe
    // idle (the next MFMA of an in-order wave cannot issue until they are out).
    constexpr int Q = 4 * MT * NT;               // MFMAs per k-step
    auto mf = [&](const f32x4 (&fa)[MT], const f32x4 (&fb)[NT], int idx) {
        const int kk = idx / (MT * NT), i = (idx / NT) % MT, j = idx % NT;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
    };
    auto store_row = [&](const f32x4 (&ar)[AI], const f32x4 (&br)[BI], int buf, int r) {
        if (r < AI) *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32 * r) * LDK + lcol * 4]) = ar[r];
        else *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + 32 * (r - AI)) * LDK + lcol * 4]) = br[r - AI];
    };
    auto chunk = [&](int t, int buf, f32x4 (&lda)[AI], f32x4 (&ldb)[BI], const f32x4 (&sta)[AI],
                     const f32x4 (&stb)[BI]) {
        // (dev ablation flags in V: 1 no A traffic, 2 no LDS writes, 4 no LDS reads, 8 no barrier,
        //  16 A from one line, 32 B from one line)
        // k-step 0 (+ the A gathers of chunk t+2)
        if (!(V & 4)) read_frags(fa1, fb1, buf, 1);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            mf(fa0, fb0, q);
            if (q % (Q / AI) == 1) load_a(lda, q / (Q / AI));
        }
        // k-step 1 (+ the weight rows of chunk t+2)
        if (!(V & 4)) read_frags(fa0, fb0, buf, 2);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            mf(fa1, fb1, q);
            if (q % (Q / BI) == 1) load_b(ldb, q / (Q / BI));
            if (q == Q - 2) advance_k();
        }
        // k-step 2 (+ chunk t+1 goes to the other LDS buffer, one row per MFMA)
        if (!(V & 4)) read_frags(fa1, fb1, buf, 3);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            mf(fa0, fb0, q);
            if (q < AI + BI && !(V & 2)) store_row(sta, stb, buf ^ 1, q);
        }
        if (!(V & 8)) __syncthreads();
        if (!(V & 4)) read_frags(fa0, fb0, buf ^ 1, 0);
        // k-step 3
#pragma unroll
        for (int q = 0; q < Q; ++q) mf(fa1, fb1, q);
    };

    load_chunk(areg0, breg0);                       // chunk 0
    store_chunk(areg0, breg0, 0);
    load_chunk(areg1, breg1);                       // chunk 1 stays in registers until chunk 0's k-step 2
    __syncthreads();
    read_frags(fa0, fb0, 0, 0);
    if (V & 4) read_frags(fa1, fb1, 0, 1);

    for (int t = 0; t < nk; t += 2) {
        chunk(t, 0, areg0, breg0, areg1, breg1);
        if (t + 1 < nk) chunk(t + 1, 1, areg1, breg1, areg0, breg0);
    }

    // ---- epilogue: y = act(acc*scale + shift + residual) --------------------------------------
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // Branch-free: residual loads / stores are raw buffer ops rebased at this tile's first row;
    // rows >= M or cols >= K get an out-of-range offset (loads return 0, stores are dropped), so all
    // residual loads of a wave are in flight together instead of one vmcnt(0) per element.
    const size_t tile_off = (size_t)m0 * p.K;
    const unsigned y_bytes = (unsigned)min((size_t)BM, (size_t)(p.M - m0)) * (unsigned)p.K * 4u;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + tile_off), 0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.res ? p.res : p.y) + tile_off), 0, (int)y_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const bool nok = n < p.K;
        const float sc = (p.scale && nok) ? p.scale[n] : 1.f;
        const float sh = (p.shift && nok) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            unsigned off[16];
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                off[r] = nok ? (unsigned)(row * p.K + n) * 4u : kOob;      // rows >= M fall outside y_bytes
            }
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off[r], 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] * sc + sh;
                if (p.res) v += rv[r];
                if (p.relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, off[r], 0, 0);
            }
        }
    }
}

// OIHW [K,Cin,R,S] -> tile-contiguous packed weights [rows/BN][kred/32][BN][32] (zero fill).
// Chunk q of the reduction covers, for Cstore >= 32: tap row r = q / (S*Cstore/32), channel chunk
// cc = (q / S) % (Cstore/32), tap column s = q % S, channels cc*32 .. cc*32+31;
// for Cstore < 32 (stem): reduction index kk = q*32 + col in plain (r, s, c) order.
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int K, int Cin,
                                   int R, int S, int Cstore, int kred, int rows, int bn, int mode) {
    const size_t total = (size_t)rows * kred;
    const int nk = kred / BK;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % BK);
        size_t rest = i / BK;
        const int row_in = (int)(rest % bn);
        rest /= bn;
        const int q = (int)(rest % nk);
        const int tn = (int)(rest / nk);
        const int row = tn * bn + row_in;
        int r, s2, c;
        if (Cstore >= BK) {
            const int ccn = Cstore / BK;
            s2 = q % S;
            c = ((q / S) % ccn) * BK + col;
            r = q / (S * ccn);
        } else {
            const int kk = q * BK + col;
            const int pos = kk / Cstore;
            c = kk - pos * Cstore;
            r = pos / S;
            s2 = pos - r * S;
        }
        float v = 0.f;
        if (row < K && c < Cin && r < R) {
            if (mode == 0) {
                v = w[(((size_t)row * Cin + c) * R + r) * S + s2];
            } else {  // ConvTranspose2d weight [Cin, Cout, 2, 2]; row = (a*2+b)*Cout + co
                const int cout = K / 4;
                const int ab = row / cout;
                const int co = row - ab * cout;
                v = w[(((size_t)c * cout + co) * 2 + (ab >> 1)) * 2 + (ab & 1)];
            }
        }
        out[i] = v;
    }
}

}  // namespace

extern "C" {

int seam_conv_kred(int C, int R, int S) { return ((R * S * C + BK - 1) / BK) * BK; }
int seam_conv_rows_padded(int K) { return ((K + 63) / 64) * 64; }

int seam_pack_conv_weight_f32(const float* w, float* w_packed, int K, int Cin, int R, int S, int Cstore,
                              int mode, void* stream) {
    const int kred = seam_conv_kred(Cstore, R, S);
    const int rows = seam_conv_rows_padded(K);
    const size_t total = (size_t)rows * kred;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (Cstore >= BK && Cstore % BK) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, w_packed, K, Cin, R,
                       S, Cstore, kred, rows, rows % 128 == 0 ? 128 : 64, mode);
    return (int)hipGetLastError();
}

int seam_conv2d_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                    const float* residual, float* y, int N, int H, int W, int C, int K, int R, int S, int stride,
                    int pad, int relu, void* stream) {
    if ((C & 3) || (C >= BK && C % BK) || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
    ConvArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.H = H; a.W = W; a.C = C;
    a.Ho = (H + 2 * pad - R) / stride + 1;
    a.Wo = (W + 2 * pad - S) / stride + 1;
    a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
    a.kred = seam_conv_kred(C, R, S);
    a.M = N * a.Ho * a.Wo;
    a.relu = relu;
    if (a.Ho <= 0 || a.Wo <= 0) return (int)hipErrorInvalidValue;
    const int rows = seam_conv_rows_padded(K);
    static const int variant = getenv("SEAM_CONV_VARIANT") ? atoi(getenv("SEAM_CONV_VARIANT")) : 0;   // dev knob
    if (rows % 128 == 0) {
        a.tiles_m = (a.M + 127) / 128;
        a.tiles_n = rows / 128;
        const dim3 g(a.tiles_m * a.tiles_n), b(256);
        if (variant == 1) hipLaunchKernelGGL((conv_igemm_f32<128, 128, 1>), g, b, 0, (hipStream_t)stream, a);
        else if (variant == 16) hipLaunchKernelGGL((conv_igemm_f32<128, 128, 16>), g, b, 0, (hipStream_t)stream, a);
        else if (variant == 32) hipLaunchKernelGGL((conv_igemm_f32<128, 128, 32>), g, b, 0, (hipStream_t)stream, a);
        else if (variant == 48) hipLaunchKernelGGL((conv_igemm_f32<128, 128, 48>), g, b, 0, (hipStream_t)stream, a);
        else if (variant == 17) hipLaunchKernelGGL((conv_igemm_f32<128, 128, 17>), g, b, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((conv_igemm_f32<128, 128>), g, b, 0, (hipStream_t)stream, a);
    } else {
        a.tiles_m = (a.M + 127) / 128;
        a.tiles_n = rows / 64;
        hipLaunchKernelGGL((conv_igemm_f32<128, 64>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0,
                           (hipStream_t)stream, a);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
