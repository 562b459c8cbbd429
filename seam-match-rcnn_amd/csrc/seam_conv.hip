// seam_conv.hip -- implicit-GEMM convolution on gfx950, exact fp32 on the matrix cores.
//
// One kernel family serves every dense contraction of the path (ResNet-50 body, FPN, RPN head,
// box/mask heads, the match trunk's valid 3x3 convs and its Linear) -- see include/seam_hip.h.
//
// GEMM view (TN):  Y[M, K] = A[M, kred] * B[K, kred]^T
//   M    = N*Ho*Wo output pixels, row m -> (n, ho, wo)          (NHWC output == row-major [M][K])
//   kred = (r, s, c) with c fastest; A is gathered on the fly from the NHWC input (zero fill for
//          padding / tails), B = pre-packed weights, K-contiguous rows.
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

template <int BM, int BN>
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
    const int lcol = tid & 7;     // which float4 of the 32-wide k chunk
    const int lrow = tid >> 3;    // 0..31
    const float* abase[AI];
    int ahi[AI], awi[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            const int n = m / (p.Ho * p.Wo);
            const int rm = m - n * (p.Ho * p.Wo);
            const int ho = rm / p.Wo;
            const int wo = rm - ho * p.Wo;
            abase[i] = p.x + (size_t)n * p.H * p.W * p.C;
            ahi[i] = ho * p.stride - p.pad;
            awi[i] = wo * p.stride - p.pad;
        } else {
            abase[i] = p.x;
            ahi[i] = -(1 << 28);
            awi[i] = 0;
        }
    }
    const float* bptr[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) bptr[i] = p.w + (size_t)(n0 + lrow + 32 * i) * p.kred + lcol * 4;

    // (r, s, c) of this thread's float4 inside the current chunk
    int kc, kr, ks;
    {
        const int kk = lcol * 4;
        const int pos = kk / p.C;
        kc = kk - pos * p.C;
        kr = pos / p.S;
        ks = pos - kr * p.S;
    }

    f32x4 areg[AI], breg[BI];
    unsigned aok = 0;   // per-row validity of the chunk held in areg (applied when it is written to LDS)
    auto load_chunk = [&]() {
        // Loads are unconditional (an out-of-image tap reads a safe in-bounds address and is zeroed at
        // the LDS write): a branch around a load makes hipcc serialise the chunk behind vmcnt waits.
        aok = 0;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int hi = ahi[i] + kr;
            const int wi = awi[i] + ks;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && kr < p.R;
            const size_t off = ok ? ((size_t)(hi * p.W + wi) * p.C + kc) : (size_t)0;
            areg[i] = *reinterpret_cast<const f32x4*>(abase[i] + off);
            aok |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            breg[i] = *reinterpret_cast<const f32x4*>(bptr[i]);
            bptr[i] += BK;
        }
        // advance (r,s,c) by one chunk
        kc += BK;
        if (p.C >= BK) {
            if (kc >= p.C) {
                kc -= p.C;
                if (++ks == p.S) { ks = 0; ++kr; }
            }
        } else {
            while (kc >= p.C) {
                kc -= p.C;
                if (++ks == p.S) { ks = 0; ++kr; }
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32 * i) * LDK + lcol * 4]) = ((aok >> i) & 1u) ? areg[i] : z;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i)
            *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + 32 * i) * LDK + lcol * 4]) = breg[i];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.kred / BK;
    load_chunk();
    store_chunk(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fk = (lane >> 5) * 4;

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) load_chunk();           // global loads in flight under the MFMAs below
        const float* as = &As[buf][(wm0 + frow) * LDK + fk];
        const float* bs = &Bs[buf][(wn0 + frow) * LDK + fk];
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDK + k8 * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDK + k8 * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nk) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: y = act(acc*scale + shift + residual) --------------------------------------
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const bool nok = n < p.K;
        const float sc = (p.scale && nok) ? p.scale[n] : 1.f;
        const float sh = (p.shift && nok) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (nok && m < p.M) {
                    const size_t o = (size_t)m * p.K + n;
                    float v = acc[i][j][r] * sc + sh;
                    if (p.res) v += p.res[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[o] = v;
                }
            }
        }
    }
}

// OIHW [K,Cin,R,S] -> [rows_padded][kred], reduction index (r,s,c), zero fill.
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int K, int Cin,
                                   int R, int S, int Cstore, int kred, int rows, int mode) {
    const size_t total = (size_t)rows * kred;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / kred);
        const int kk = (int)(i - (size_t)row * kred);
        const int pos = kk / Cstore;
        const int c = kk - pos * Cstore;
        const int r = pos / S;
        const int s = pos - r * S;
        float v = 0.f;
        if (row < K && c < Cin && r < R) {
            if (mode == 0) {
                v = w[(((size_t)row * Cin + c) * R + r) * S + s];
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
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, w_packed, K, Cin, R,
                       S, Cstore, kred, rows, mode);
    return (int)hipGetLastError();
}

int seam_conv2d_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                    const float* residual, float* y, int N, int H, int W, int C, int K, int R, int S, int stride,
                    int pad, int relu, void* stream) {
    if ((C & 3) || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
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
    if (rows % 128 == 0) {
        a.tiles_m = (a.M + 127) / 128;
        a.tiles_n = rows / 128;
        hipLaunchKernelGGL((conv_igemm_f32<128, 128>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0,
                           (hipStream_t)stream, a);
    } else {
        a.tiles_m = (a.M + 127) / 128;
        a.tiles_n = rows / 64;
        hipLaunchKernelGGL((conv_igemm_f32<128, 64>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0,
                           (hipStream_t)stream, a);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
