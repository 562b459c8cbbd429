// seam_heads.hip -- SEAM temporal head kernels for gfx950 (wave64):
//   nlb_attnpool  concatenation-form non-local block + softmax attention pooling, one workgroup
//                 per sequence, all intermediates on chip (reference: ~12 launches + a
//                 [1,256,T,T] temporary PER SEQUENCE inside a Python loop)
//   pair_logits   x5[i,j,:] = W * (a_i - b_j)^2 + bias, register-blocked direct form (the direct
//                 form is the parity reference; no a^2+b^2-2ab cancellation)
//   rank_topk     descending rank of softmax(x5)[...,1], k rounds of a workgroup arg-max
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "seam_topk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

using seam_topk::TopkShared;
using seam_topk::block_topk;

constexpr int D = 256;       // descriptor width
constexpr int DI = 128;      // NLB inter channels
constexpr int RC = 16;       // rows per chunk (a 10-frame sequence is one pass over the projection weights)
constexpr int T_LDS = 96;    // sequences up to this length keep G/a/b in LDS

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct NlbArgs {
    const float* seq;
    int64_t t_stride, s_stride;
    const int* len;
    int S, Tmax;
    const float* w_proj_t;   // [256][384]
    const float* b_proj;     // [384]
    const float* w_cat;      // [256]
    const float* w_out_t;    // [128][256]
    const float* b_out;      // [256]
    const float* w_att;      // [256]
    const float* b_att;      // [1]
    float* out;              // [S][256]
    float* att;              // [S][Tmax] or null
    float* z;                // [S][Tmax][256] or null: the block output Z (T==1: X)
    float* ws;               // per sequence Tmax*(128+2) floats
    int use_nlb;
};

// dynamic LDS layout (floats): xs[RC][256] | ys[RC][128] | red[RC][4] | sc[RC] | G[T_LDS][128] | a[T_LDS] | b[T_LDS]
__global__ __launch_bounds__(256) void nlb_attnpool_kernel(const NlbArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                       // RC*256
    float* ys = xs + RC * D;                // RC*128
    float* red = ys + RC * DI;              // RC*4
    float* scs = red + RC * 4;              // RC
    float* gl = scs + RC;                   // LDS home of G/a/b

    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int T = min(p.len[s], p.Tmax);              // a length beyond the packed rows would index LDS / scratch out of range
    const float* X = p.seq + (int64_t)s * p.s_stride;
    float* outp = p.out + (size_t)s * D;

    if (T <= 0) { outp[tid] = 0.f; return; }

    const bool nlb = p.use_nlb == 2 || (p.use_nlb && T > 1);   // 2 = apply even to a single row
    float* G;
    float* av;
    float* bv;
    if (T <= T_LDS) {
        G = gl; av = gl + T_LDS * DI; bv = av + T_LDS;
    } else {
        float* w = p.ws + (size_t)s * p.Tmax * (DI + 2);
        G = w; av = w + (size_t)p.Tmax * DI; bv = av + p.Tmax;
    }

    if (nlb) {
        // ---- phase 1: TH/PH/G projections, a = TH.wc[:128], b = PH.wc[128:] -------------------
        const float wc = p.w_cat[tid];                 // thread tid owns theta col tid (<128) / phi col tid-128
        const float bp0 = p.b_proj[tid];
        const float bp1 = tid < DI ? p.b_proj[256 + tid] : 0.f;
        for (int r0 = 0; r0 < T; r0 += RC) {
            const int nr = min(RC, T - r0);
            __syncthreads();
            for (int r = 0; r < RC; ++r) xs[r * D + tid] = r < nr ? X[(int64_t)(r0 + r) * p.t_stride + tid] : 0.f;
            __syncthreads();
            float a0[RC], a1[RC];
#pragma unroll
            for (int r = 0; r < RC; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
            for (int k = 0; k < D; ++k) {
                const float w0 = p.w_proj_t[(size_t)k * 384 + tid];
                const float w1 = tid < DI ? p.w_proj_t[(size_t)k * 384 + 256 + tid] : 0.f;
#pragma unroll
                for (int r = 0; r < RC; ++r) {
                    const float xv = xs[r * D + k];
                    a0[r] = fmaf(xv, w0, a0[r]);
                    a1[r] = fmaf(xv, w1, a1[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < RC; ++r) {
                const float v = wave_sum((a0[r] + bp0) * wc);     // waves 0,1: theta ; waves 2,3: phi
                if (lane == 0) red[r * 4 + wid] = v;
                if (tid < DI && r < nr) G[(size_t)(r0 + r) * DI + tid] = a1[r] + bp1;
            }
            __syncthreads();
            if (tid < RC && tid < nr) {
                av[r0 + tid] = red[tid * 4 + 0] + red[tid * 4 + 1];
                bv[r0 + tid] = red[tid * 4 + 2] + red[tid * 4 + 3];
            }
        }
        __syncthreads();
    }

    // ---- phase 2: Y = f G ; Z = Y Ww^T + bw + X ; online-softmax attention pooling ----------------
    const float wa = p.w_att[tid];
    const float ba = p.b_att[0];
    const float bo = p.b_out[tid];
    float m_run = -INFINITY, l_run = 0.f, o_run = 0.f;
    const float Tf = (float)T;
    for (int r0 = 0; r0 < T; r0 += RC) {
        const int nr = min(RC, T - r0);
        float z[RC];
        if (nlb) {
            {   // Y rows: thread (c = tid&127, half = tid>>7) -> rows half, half+2, half+4, half+6
                const int c = tid & (DI - 1), hf = tid >> 7;
                float ai[RC / 2], y[RC / 2];
#pragma unroll
                for (int q = 0; q < RC / 2; ++q) {
                    const int r = hf + 2 * q;
                    ai[q] = r < nr ? av[r0 + r] : 0.f;
                    y[q] = 0.f;
                }
                for (int j = 0; j < T; ++j) {
                    const float bj = bv[j];
                    const float g = G[(size_t)j * DI + c];
#pragma unroll
                    for (int q = 0; q < RC / 2; ++q) y[q] = fmaf(fmaxf(ai[q] + bj, 0.f) / Tf, g, y[q]);
                }
                __syncthreads();   // previous chunk's readers of ys are done
#pragma unroll
                for (int q = 0; q < RC / 2; ++q) ys[(hf + 2 * q) * DI + c] = y[q];
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < RC; ++r) z[r] = 0.f;
            for (int c = 0; c < DI; ++c) {
                const float w = p.w_out_t[(size_t)c * D + tid];
#pragma unroll
                for (int r = 0; r < RC; ++r) z[r] = fmaf(ys[r * DI + c], w, z[r]);
            }
#pragma unroll
            for (int r = 0; r < RC; ++r)
                z[r] = r < nr ? z[r] + bo + X[(int64_t)(r0 + r) * p.t_stride + tid] : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < RC; ++r) z[r] = r < nr ? X[(int64_t)(r0 + r) * p.t_stride + tid] : 0.f;
        }
        if (p.z) {
#pragma unroll
            for (int r = 0; r < RC; ++r)
                if (r < nr) p.z[((size_t)s * p.Tmax + r0 + r) * D + tid] = z[r];
        }
        // scores s_r = Z_r . wa + ba
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const float v = wave_sum(z[r] * wa);
            if (lane == 0) red[r * 4 + wid] = v;
        }
        __syncthreads();
        if (tid < RC) {
            const float sv = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3] + ba;
            scs[tid] = sv;
            if (p.att && tid < nr) p.att[(size_t)s * p.Tmax + r0 + tid] = sv;   // raw score, normalised below
        }
        __syncthreads();
        float m_new = m_run;
        for (int r = 0; r < nr; ++r) m_new = fmaxf(m_new, scs[r]);
        const float corr = expf(m_run - m_new);     // exp(-inf) = 0 on the first chunk
        l_run *= corr;
        o_run *= corr;
        for (int r = 0; r < nr; ++r) {
            const float e = expf(scs[r] - m_new);
            l_run += e;
            o_run = fmaf(e, z[r], o_run);
        }
        m_run = m_new;
    }
    outp[tid] = o_run / l_run;
    if (p.att) {
        __syncthreads();
        for (int t = tid; t < T; t += 256) {
            float* q = p.att + (size_t)s * p.Tmax + t;
            *q = expf(*q - m_run) / l_run;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same block with its GEMMs on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32), one workgroup (4 waves) per
// sequence, rows in tiles of 32 (sequences up to NLB_TMF = 96 rows; longer ones take the kernel above):
//   G  = X Wg + bg            [32 x 256] x [256 x 128]   wave w -> output columns 32 w .. 32 w + 31
//   Y  = f G, f_ij = relu(a_i + b_j) / T   [32 x T] x [T x 128]   A fragments (f) are made in registers from a_i, b_j
//   Z  = Y Ww^T + bw + X      [32 x 128] x [128 x 256]   wave w -> columns 64 w .. 64 w + 63
// theta / phi enter the block only through a_i = theta_i . wc[:128] and b_j = phi_j . wc[128:] (the concat_project weight),
// so the two 256 -> 128 projections collapse to two 256-vectors u = W_theta^T wc[:128], v = W_phi^T wc[128:] (folded once
// at pack time in fp64) and a_i = X_i . u + c: a third of the block's FLOPs instead of three quarters of them.
// Weights arrive in MFMA B-fragment order ([n_tile][k / 8][lane][4], element e of lane (h = lane >> 5, n = lane & 31)
// = W[k = 8 j + 4 h + e][n]): a wave streams them with one coalesced 1 KiB load per four MFMAs; A fragments are 16-byte
// LDS reads of row-major tiles with an odd 16-byte-slot pitch (conflict-free), using the same k pairing {e, e + 4}.
constexpr int NLB_TMF = 96;
constexpr int XLD = 260, GLD = 132, YLD = 132;          // LDS pitches (floats): 65 / 33 / 33 sixteen-byte slots, odd

struct NlbMfArgs {
    const float* seq;
    int64_t t_stride, s_stride;
    const int* len;
    int S, Tmax;
    const float* wg_frag;    // [4][32][64][4]   G projection, fragment order
    const float* b_g;        // [128]
    const float* u;          // [256]  W_theta^T wc[:128]
    const float* v;          // [256]  W_phi^T   wc[128:]
    const float* cd;         // [2]    b_theta . wc[:128],  b_phi . wc[128:]
    const float* wo_frag;    // [8][16][64][4]   W (output) projection, fragment order
    const float* b_out;      // [256]
    const float* w_att;      // [256]
    const float* b_att;      // [1]
    float* out;              // [S][256]
    float* att;              // [S][Tmax] or null
    float* z;                // [S][Tmax][256] or null
    int use_nlb;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void nlb_attnpool_mfma_kernel(const NlbMfArgs p) {
    __shared__ __attribute__((aligned(16))) float Xs[32 * XLD];
    __shared__ __attribute__((aligned(16))) float Gs[NLB_TMF * GLD];
    __shared__ __attribute__((aligned(16))) float Ys[32 * YLD];
    __shared__ float av[NLB_TMF], bv[NLB_TMF], red[32 * 4], scs[32];

    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int nl = lane & 31, h = lane >> 5;
    const int T = min(p.len[s], p.Tmax);              // a length beyond the packed rows would index LDS / scratch out of range
    const float* X = p.seq + (int64_t)s * p.s_stride;
    float* outp = p.out + (size_t)s * D;
    if (T <= 0) { outp[tid] = 0.f; return; }
    const bool nlb = p.use_nlb == 2 || (p.use_nlb && T > 1);   // 2 = apply even to a single row
    const int ntile = (T + 31) >> 5;

    auto load_x_tile = [&](int r0) {       // rows r0 .. r0 + 31 of X -> Xs (zeros past the end); 8 x 16 bytes per thread
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = w + 4 * i, c4 = lane;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (r0 + row < T) x = *reinterpret_cast<const f32x4*>(X + (int64_t)(r0 + row) * p.t_stride + c4 * 4);
            *reinterpret_cast<f32x4*>(&Xs[row * XLD + c4 * 4]) = x;
        }
    };

    if (nlb) {
        // ---- phase 1: a, b and G for every row -------------------------------------------------------------------
        const f32x4 u4 = *reinterpret_cast<const f32x4*>(p.u + lane * 4), v4 = *reinterpret_cast<const f32x4*>(p.v + lane * 4);
        const float c0 = p.cd[0], d0 = p.cd[1];
        const float bg = p.b_g[32 * w + nl];
        for (int t = 0; t < ntile; ++t) {
            const int r0 = 32 * t;
            __syncthreads();                       // readers of the previous tile's Xs are done
            load_x_tile(r0);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {          // wave w: rows 8 w .. 8 w + 7
                const int row = 8 * w + i;
                const f32x4 x = *reinterpret_cast<const f32x4*>(&Xs[row * XLD + lane * 4]);
                const float pa = wave_sum(x[0] * u4[0] + x[1] * u4[1] + x[2] * u4[2] + x[3] * u4[3]);
                const float pb = wave_sum(x[0] * v4[0] + x[1] * v4[1] + x[2] * v4[2] + x[3] * v4[3]);
                if (lane == 0 && r0 + row < NLB_TMF) { av[r0 + row] = pa + c0; bv[r0 + row] = pb + d0; }
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const f32x4* bf = reinterpret_cast<const f32x4*>(p.wg_frag) + (size_t)w * 32 * 64 + lane;
#pragma unroll 4
            for (int j = 0; j < 32; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&Xs[nl * XLD + 8 * j + 4 * h]);
                const f32x4 b = bf[j * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < NLB_TMF) Gs[row * GLD + 32 * w + nl] = acc[r] + bg;
            }
        }
        __syncthreads();
    }

    // ---- phase 2: Y = f G ; Z = Y Ww^T + bw + X ; online-softmax attention pooling --------------------------------
    const int col0 = 64 * w + nl, col1 = col0 + 32;      // this lane's two output columns (C layout: col = lane & 31 per n-tile)
    const float wa0 = p.w_att[col0], wa1 = p.w_att[col1];
    const float bo0 = p.b_out[col0], bo1 = p.b_out[col1];
    const float ba = p.b_att[0];
    const float inv_t = 1.f / (float)T;
    float m_run = -INFINITY, l_run = 0.f, o0 = 0.f, o1 = 0.f;
    const int kpad = (T + 7) & ~7;
    for (int t = 0; t < ntile; ++t) {
        const int r0 = 32 * t;
        const int nr = min(32, T - r0);
        if (!(nlb && ntile == 1)) {                // (one-tile sequences still hold their X tile from phase 1)
            __syncthreads();
            load_x_tile(r0);
        }
        float z0[16], z1[16];
        if (nlb) {
            {   // Y tile: wave w -> columns 32 w .. 32 w + 31
                const float ai = av[min(r0 + nl, NLB_TMF - 1)];
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                for (int jj = 0; jj < kpad; jj += 8) {
                    f32x4 a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = jj + 4 * h + e;
                        const bool ok = k < T;
                        a[e] = ok ? fmaxf(ai + bv[ok ? k : 0], 0.f) * inv_t : 0.f;
                        b[e] = ok ? Gs[k * GLD + 32 * w + nl] : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
                }
                __syncthreads();                   // previous tile's readers of Ys are done; Xs of this tile is complete
#pragma unroll
                for (int r = 0; r < 16; ++r) Ys[((r & 3) + 8 * (r >> 2) + 4 * h) * YLD + 32 * w + nl] = acc[r];
                __syncthreads();
            }
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            const f32x4* bf = reinterpret_cast<const f32x4*>(p.wo_frag) + (size_t)(2 * w) * 16 * 64 + lane;
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&Ys[nl * YLD + 8 * j + 4 * h]);
                const f32x4 b0 = bf[j * 64], b1 = bf[(16 + j) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b1[e], acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const bool ok = row < nr;
                z0[r] = ok ? acc0[r] + bo0 + Xs[row * XLD + col0] : 0.f;
                z1[r] = ok ? acc1[r] + bo1 + Xs[row * XLD + col1] : 0.f;
            }
        } else {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const bool ok = row < nr;
                z0[r] = ok ? Xs[row * XLD + col0] : 0.f;
                z1[r] = ok ? Xs[row * XLD + col1] : 0.f;
            }
        }
        if (p.z) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < nr) {
                    float* q = p.z + ((size_t)s * p.Tmax + r0 + row) * D;
                    q[col0] = z0[r];
                    q[col1] = z1[r];
                }
            }
        }
        // scores s_row = Z_row . wa + ba: a lane holds 2 of a row's 256 columns; reduce over the 32 lanes of its half,
        // then over the 4 waves through LDS
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = z0[r] * wa0 + z1[r] * wa1;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (nl == 0) red[((r & 3) + 8 * (r >> 2) + 4 * h) * 4 + w] = v;
        }
        __syncthreads();
        if (tid < 32) {
            const float sv = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3] + ba;
            scs[tid] = sv;
            if (p.att && tid < nr) p.att[(size_t)s * p.Tmax + r0 + tid] = sv;   // raw score, normalised below
        }
        __syncthreads();
        float m_new = m_run;
        for (int r = 0; r < nr; ++r) m_new = fmaxf(m_new, scs[r]);
        const float corr = expf(m_run - m_new);     // exp(-inf) = 0 on the first tile
        l_run *= corr;
        o0 *= corr;
        o1 *= corr;
        for (int r = 0; r < nr; ++r) l_run += expf(scs[r] - m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float e = row < nr ? expf(scs[row] - m_new) : 0.f;
            o0 = fmaf(e, z0[r], o0);
            o1 = fmaf(e, z1[r], o1);
        }
        m_run = m_new;
    }
    o0 += __shfl_xor(o0, 32, 64);                   // the two row halves of the C layout
    o1 += __shfl_xor(o1, 32, 64);
    if (h == 0) {
        outp[col0] = o0 / l_run;
        outp[col1] = o1 / l_run;
    }
    if (p.att) {
        __syncthreads();
        for (int t = tid; t < T; t += 256) {
            float* q = p.att + (size_t)s * p.Tmax + t;
            *q = expf(*q - m_run) / l_run;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pair logits: block tile (8*QT) x (32*GT) pairs, thread tile QT x GT, k chunks of 32 through LDS.
// The two classes of a pair live in one float2 so the inner loop is packed fp32 math
// (v_pk_add / v_pk_mul / v_pk_fma: the fp32 vector peak needs the packed forms): per k and per
// (query, 2 products) one packed subtract + one packed square, per pair one packed fma into (x0, x1).
// Same operation order per pair as the scalar form (sub, mul, fma chain over k) => bit-identical logits.
// Chunks are software pipelined: next chunk global -> registers while the current one is consumed from
// LDS, double-buffered LDS, one barrier per chunk.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int PKC = 32, PLD = PKC + 4;

template <int QT, int GT>
struct PairLds {
    float as[2][8 * QT * PLD];
    float bs[2][32 * GT * PLD];
    float ws[2][2 * PKC];          // interleaved (w0[k], w1[k])
};

template <int QT, int GT>
__device__ __forceinline__ void pair_tile(const float* __restrict__ a, const float* __restrict__ b,
                                          const float* __restrict__ w, int q0, int g0, int Q, int G, int Dd,
                                          PairLds<QT, GT>& L, f32x2 (&acc)[QT][GT]) {
    constexpr int BQ = 8 * QT, BG = 32 * GT;
    constexpr int NA = (BQ * (PKC / 4) + 255) / 256, NB = (BG * (PKC / 4)) / 256;
    static_assert(GT % 2 == 0 && (BG * (PKC / 4)) % 256 == 0, "tile shape");
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    f32x4 ra[NA], rb[NB];
    float rw = 0.f;
    auto gload = [&](int k0) {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int i = tid + 256 * t, r = i / (PKC / 4), c = i % (PKC / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < BQ * (PKC / 4) && q0 + r < Q) v = *reinterpret_cast<const f32x4*>(a + (size_t)(q0 + r) * Dd + k0 + c * 4);
            ra[t] = v;
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int i = tid + 256 * t, r = i / (PKC / 4), c = i % (PKC / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g0 + r < G) v = *reinterpret_cast<const f32x4*>(b + (size_t)(g0 + r) * Dd + k0 + c * 4);
            rb[t] = v;
        }
        if (tid < 2 * PKC) rw = w[(size_t)(tid & 1) * Dd + k0 + (tid >> 1)];
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int i = tid + 256 * t, r = i / (PKC / 4), c = i % (PKC / 4);
            if (i < BQ * (PKC / 4)) *reinterpret_cast<f32x4*>(&L.as[buf][r * PLD + c * 4]) = ra[t];
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int i = tid + 256 * t, r = i / (PKC / 4), c = i % (PKC / 4);
            *reinterpret_cast<f32x4*>(&L.bs[buf][r * PLD + c * 4]) = rb[t];
        }
        if (tid < 2 * PKC) L.ws[buf][tid] = rw;
    };
#pragma unroll
    for (int i = 0; i < QT; ++i)
#pragma unroll
        for (int j = 0; j < GT; ++j) acc[i][j] = (f32x2){0.f, 0.f};

    gload(0);
    __syncthreads();                 // previous user of the LDS buffers is done
    lstore(0);
    __syncthreads();
    const int nch = Dd / PKC;
    for (int ch = 0; ch < nch; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nch) gload((ch + 1) * PKC);
        const float* as = L.as[buf];
        const float* bs = L.bs[buf];
        const float* ws = L.ws[buf];
#pragma unroll
        for (int k4 = 0; k4 < PKC; k4 += 4) {
            f32x4 av[QT], bv[GT];
#pragma unroll
            for (int i = 0; i < QT; ++i) av[i] = *reinterpret_cast<const f32x4*>(&as[(ty * QT + i) * PLD + k4]);
#pragma unroll
            for (int j = 0; j < GT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(&bs[(tx + 32 * j) * PLD + k4]);
            f32x2 wk[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wk[kk] = *reinterpret_cast<const f32x2*>(&ws[(k4 + kk) * 2]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < QT; ++i)
#pragma unroll
                    for (int j = 0; j < GT; j += 2) {
                        const f32x2 aa = {av[i][kk], av[i][kk]};
                        const f32x2 bb = {bv[j][kk], bv[j + 1][kk]};
                        const f32x2 d = aa - bb;
                        const f32x2 d2 = d * d;
                        acc[i][j] = __builtin_elementwise_fma((f32x2){d2.x, d2.x}, wk[kk], acc[i][j]);
                        acc[i][j + 1] = __builtin_elementwise_fma((f32x2){d2.y, d2.y}, wk[kk], acc[i][j + 1]);
                    }
        }
        if (ch + 1 < nch) lstore(buf ^ 1);
        __syncthreads();
    }
}

template <int QT, int GT>
__global__ __launch_bounds__(256) void pair_logits_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int Q, int G, int Dd) {
    constexpr int BQ = 8 * QT, BG = 32 * GT;
    __shared__ __attribute__((aligned(16))) PairLds<QT, GT> L;
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;
    const int q0 = blockIdx.y * BQ, g0 = blockIdx.x * BG;
    f32x2 acc[QT][GT];
    pair_tile<QT, GT>(a, b, w, q0, g0, Q, G, Dd, L, acc);
    const f32x2 bz = {bias[0], bias[1]};
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const int qi = q0 + ty * QT + i;
        if (qi >= Q) continue;
#pragma unroll
        for (int j = 0; j < GT; ++j) {
            const int gj = g0 + tx + 32 * j;
            if (gj < G) *reinterpret_cast<f32x2*>(out + ((size_t)qi * G + gj) * 2) = acc[i][j] + bz;
        }
    }
}

// Block-diagonal self-similarity (evaluator tracking, ref evaluate_movingfashion.py:165-176 `compute_selfdist` per product): for every
// segment s of rows [seg[s], seg[s+1]) of x, score[out_off[s] + i * n + j] = softmax(W (x_i - x_j)^2 + b)[1] for all pairs INSIDE the
// segment -- the n_s x n_s diagonal blocks of the all-pairs matrix only, in one launch.  Same tile, FMA order, bias add and softmax
// expression as pair_logits_kernel<2,2> + match_scores_kernel: bit-identical to computing each block on its own.
// grid (max tiles of a segment, segments); blocks past a segment's tile count exit.
__global__ __launch_bounds__(256) void pair_scores_blockdiag_kernel(const float* __restrict__ x, const int* __restrict__ seg,
                                                                    const int64_t* __restrict__ out_off, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, float* __restrict__ out, int Dd) {
    constexpr int QT = 2, GT = 2, BQ = 8 * QT, BG = 32 * GT;
    __shared__ __attribute__((aligned(16))) PairLds<QT, GT> L;
    const int s = blockIdx.y;
    const int r0 = seg[s], n = seg[s + 1] - r0;
    const int tg_n = (n + BG - 1) / BG, tq_n = (n + BQ - 1) / BQ;
    if ((int)blockIdx.x >= tg_n * tq_n) return;
    const int tq = blockIdx.x / tg_n, tg = blockIdx.x - tq * tg_n;
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;
    const int q0 = tq * BQ, g0 = tg * BG;
    const float* xs = x + (size_t)r0 * Dd;
    f32x2 acc[QT][GT];
    pair_tile<QT, GT>(xs, xs, w, q0, g0, n, n, Dd, L, acc);
    const f32x2 bz = {bias[0], bias[1]};
    float* o = out + out_off[s];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const int qi = q0 + ty * QT + i;
        if (qi >= n) continue;
#pragma unroll
        for (int j = 0; j < GT; ++j) {
            const int gj = g0 + tx + 32 * j;
            if (gj >= n) continue;
            const f32x2 lg = acc[i][j] + bz;
            const float mx = fmaxf(lg.x, lg.y);
            const float e0 = expf(lg.x - mx), e1 = expf(lg.y - mx);
            o[(size_t)qi * n + gj] = e1 / (e0 + e1);
        }
    }
}

__global__ __launch_bounds__(256) void rank_topk_kernel(const float* __restrict__ logits, int64_t* __restrict__ idx,
                                                        float* __restrict__ score, int G, int k) {
    __shared__ TopkShared sh;
    const int q = blockIdx.x;
    const float2* row = reinterpret_cast<const float2*>(logits) + (size_t)q * G;
    block_topk([&](int j, float& x0, float& x1, int& g) { const float2 x = row[j]; x0 = x.x; x1 = x.y; g = j; }, G, k,
               idx + (size_t)q * k, score + (size_t)q * k, sh);
}

// ------------------------------------------------------------------------------------------------
// Fused pair logits + top-k (no [Q,G,2] round trip through HBM): stage 1 computes a 32-query x
// 256-product logit tile into LDS with the same register tiling / FMA order as pair_logits_kernel<4,4>
// (so d = x1 - x0 is bit-identical to the unfused path) and selects the tile's k best per query;
// stage 2 merges the per-segment candidates.  Candidate = (x0, x1, index).
constexpr int TK_SEG = 256;

__device__ __forceinline__ bool tk_better(float d, int g, float bd, int bg) { return d > bd || (d == bd && g < bg); }

__global__ __launch_bounds__(256) void pair_topk_stage1(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ cand, int Q, int G, int Dd, int k, int nseg) {
    constexpr int BQ = 32, BG = 128;
    __shared__ __attribute__((aligned(16))) PairLds<4, 4> L;
    __shared__ float L0[BQ][TK_SEG + 1];
    __shared__ float L1[BQ][TK_SEG + 1];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5, lane = tid & 63, wid = tid >> 6;
    const int seg = blockIdx.x, q0 = blockIdx.y * BQ, gseg = seg * TK_SEG;
    const float b0 = bias[0], b1 = bias[1];
    for (int sub = 0; sub < TK_SEG / BG; ++sub) {
        const int g0 = gseg + sub * BG;
        f32x2 acc[4][4];
        pair_tile<4, 4>(a, b, w, q0, g0, Q, G, Dd, L, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                L0[ty * 4 + i][sub * BG + tx + 32 * j] = acc[i][j].x + b0;
                L1[ty * 4 + i][sub * BG + tx + 32 * j] = acc[i][j].y + b1;
            }
    }
    __syncthreads();
    // selection: wave w owns queries w*8 .. w*8+7; k rounds of wave arg-max over the 256 columns
    for (int qq = 0; qq < 8; ++qq) {
        const int ql = wid * 8 + qq, q = q0 + ql;
        if (q >= Q) break;
        float dv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = lane + 64 * c;
            float d = L1[ql][col] - L0[ql][col];
            if (d != d || gseg + col >= G) d = -INFINITY;
            dv[c] = d;
        }
        float pv = INFINITY;
        int pg = -1;
        for (int round = 0; round < k; ++round) {
            float bd = -INFINITY;
            int bg = 0x7fffffff;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int g = gseg + lane + 64 * c;
                const bool elig = g < G && ((dv[c] < pv) || (dv[c] == pv && g > pg));
                if (elig && tk_better(dv[c], g, bd, bg)) { bd = dv[c]; bg = g; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float od = __shfl_xor(bd, o, 64);
                const int og = __shfl_xor(bg, o, 64);
                if (tk_better(od, og, bd, bg)) { bd = od; bg = og; }
            }
            if (lane == 0) {
                float* c = cand + (((size_t)q * nseg + seg) * k + round) * 3;
                const bool found = bg != 0x7fffffff;
                c[0] = found ? L0[ql][bg - gseg] : 0.f;
                c[1] = found ? L1[ql][bg - gseg] : -INFINITY;
                c[2] = __int_as_float(found ? bg : -1);
            }
            pv = bd;
            pg = bg;
        }
    }
}

__global__ __launch_bounds__(256) void pair_topk_stage2(const float* __restrict__ cand, int64_t* __restrict__ idx,
                                                        float* __restrict__ score, int ncand, int k) {
    __shared__ TopkShared sh;
    const int q = blockIdx.x;
    const float* c = cand + (size_t)q * ncand * 3;
    block_topk([&](int j, float& x0, float& x1, int& g) { x0 = c[j * 3]; x1 = c[j * 3 + 1]; g = __float_as_int(c[j * 3 + 2]); },
               ncand, k, idx + (size_t)q * k, score + (size_t)q * k, sh);
}

// ------------------------------------------------------------------------------------------------
// Evaluator-side helpers: softmax(x)[...,1] of a logits matrix, and the rank of
// one target column per query under the same strict order as rank_topk (score desc, index asc).
__global__ void match_scores_kernel(const float* __restrict__ logits, float* __restrict__ score, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float2 x = reinterpret_cast<const float2*>(logits)[i];
        const float mx = fmaxf(x.x, x.y);
        const float e0 = expf(x.x - mx), e1 = expf(x.y - mx);
        score[i] = e1 / (e0 + e1);
    }
}

__global__ __launch_bounds__(256) void rank_of_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                      int64_t* __restrict__ rank, int G) {
    __shared__ int part[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const float2* row = reinterpret_cast<const float2*>(logits) + (size_t)q * G;
    const int t = (int)target[q];
    int cnt = 0;
    if (t >= 0 && t < G) {
        const float2 xt = row[t];
        float dt = xt.y - xt.x;
        if (dt != dt) dt = -INFINITY;
        for (int g = tid; g < G; g += 256) {
            const float2 x = row[g];
            float d = x.y - x.x;
            if (d != d) d = -INFINITY;
            cnt += (d > dt || (d == dt && g < t)) ? 1 : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0) part[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) rank[q] = (t >= 0 && t < G) ? (int64_t)(part[0] + part[1] + part[2] + part[3]) : (int64_t)-1;
}

// Column reduce of per-frame match scores [n,G] -> [G]: mode 0 = mean, 1 = max (the AVG / MAX DISTANCE
// rankings, ref evaluate_movingfashion.py:293-296,305).
__global__ void score_reduce_kernel(const float* __restrict__ s, float* __restrict__ out, int n, int G, int mode) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    float acc = mode ? -INFINITY : 0.f;
    for (int i = 0; i < n; ++i) {
        const float v = s[(size_t)i * G + g];
        acc = mode ? fmaxf(acc, v) : acc + v;
    }
    out[g] = mode ? acc : acc / (float)n;
}

// The same reduction over row SEGMENTS: rows seg[p] .. seg[p+1]-1 -> out[p, :] (one launch for all products of an evaluator pass;
// per column the rows are visited in the same order as score_reduce_kernel visits them: bit-identical to per-segment calls)
__global__ void score_reduce_seg_kernel(const float* __restrict__ s, const int* __restrict__ seg, float* __restrict__ out, int G, int mode) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    if (g >= G) return;
    const int lo = seg[p], hi = seg[p + 1];
    float acc = mode ? -INFINITY : 0.f;
    for (int i = lo; i < hi; ++i) {
        const float v = s[(size_t)i * G + g];
        acc = mode ? fmaxf(acc, v) : acc + v;
    }
    out[(size_t)p * G + g] = mode ? acc : acc / (float)(hi - lo);
}

// rank of target[q] in the descending order of a plain score row (ties -> lower index first)
__global__ __launch_bounds__(256) void rank_of_scores_kernel(const float* __restrict__ score, const int64_t* __restrict__ target,
                                                             int64_t* __restrict__ rank, int G) {
    __shared__ int part[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* row = score + (size_t)q * G;
    const int t = (int)target[q];
    int cnt = 0;
    if (t >= 0 && t < G) {
        float dt = row[t];
        if (dt != dt) dt = -INFINITY;
        for (int g = tid; g < G; g += 256) {
            float d = row[g];
            if (d != d) d = -INFINITY;
            cnt += (d > dt || (d == dt && g < t)) ? 1 : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0) part[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) rank[q] = (t >= 0 && t < G) ? (int64_t)(part[0] + part[1] + part[2] + part[3]) : (int64_t)-1;
}

}  // namespace

extern "C" {

int64_t seam_nlb_workspace_floats(int S, int Tmax) { return (int64_t)S * Tmax * (DI + 2) + 16; }

int seam_nlb_attnpool_f32(const float* seq, int64_t t_stride, int64_t s_stride, const int* len, int S, int Tmax,
                          const float* w_proj_t, const float* b_proj, const float* w_cat, const float* w_out_t,
                          const float* b_out, const float* w_att, const float* b_att, float* out, float* att,
                          float* z, float* ws, int use_nlb, void* stream) {
    if (S <= 0) return 0;
    NlbArgs a;
    a.seq = seq; a.t_stride = t_stride; a.s_stride = s_stride; a.len = len; a.S = S; a.Tmax = Tmax;
    a.w_proj_t = w_proj_t; a.b_proj = b_proj; a.w_cat = w_cat; a.w_out_t = w_out_t; a.b_out = b_out;
    a.w_att = w_att; a.b_att = b_att; a.out = out; a.att = att; a.z = z; a.ws = ws; a.use_nlb = use_nlb;
    const size_t lds = (size_t)(RC * D + RC * DI + RC * 4 + RC + T_LDS * DI + 2 * T_LDS) * sizeof(float);
    hipLaunchKernelGGL(nlb_attnpool_kernel, dim3(S), dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int seam_nlb_mfma_max_len(void) { return NLB_TMF; }

int seam_nlb_attnpool_mfma_f32(const float* seq, int64_t t_stride, int64_t s_stride, const int* len, int S, int Tmax,
                               const float* wg_frag, const float* b_g, const float* u, const float* v, const float* cd,
                               const float* wo_frag, const float* b_out, const float* w_att, const float* b_att, float* out,
                               float* att, float* z, int use_nlb, void* stream) {
    if (S <= 0) return 0;
    if (Tmax > NLB_TMF || (t_stride & 3) || (s_stride & 3) || ((uintptr_t)seq & 15)) return (int)hipErrorInvalidValue;
    NlbMfArgs a;
    a.seq = seq; a.t_stride = t_stride; a.s_stride = s_stride; a.len = len; a.S = S; a.Tmax = Tmax;
    a.wg_frag = wg_frag; a.b_g = b_g; a.u = u; a.v = v; a.cd = cd; a.wo_frag = wo_frag; a.b_out = b_out;
    a.w_att = w_att; a.b_att = b_att; a.out = out; a.att = att; a.z = z; a.use_nlb = use_nlb;
    hipLaunchKernelGGL(nlb_attnpool_mfma_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int seam_pair_logits_f32(const float* a, const float* b, const float* w, const float* bias, float* out, int Q,
                         int G, int Dd, void* stream) {
    if (Q <= 0 || G <= 0) return 0;
    if (Dd % 32) return (int)hipErrorInvalidValue;
    if ((long)Q * G >= (1L << 20)) {
        dim3 grid((G + 127) / 128, (Q + 31) / 32);
        hipLaunchKernelGGL((pair_logits_kernel<4, 4>), grid, dim3(256), 0, (hipStream_t)stream, a, b, w, bias, out,
                           Q, G, Dd);
    } else {
        dim3 grid((G + 63) / 64, (Q + 15) / 16);
        hipLaunchKernelGGL((pair_logits_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a, b, w, bias, out,
                           Q, G, Dd);
    }
    return (int)hipGetLastError();
}

int seam_pair_scores_blockdiag_f32(const float* x, const int* seg, const int64_t* out_off, const float* w, const float* bias,
                                   float* out, int n_seg, int max_rows, int Dd, void* stream) {
    if (n_seg <= 0 || max_rows <= 0) return 0;
    if (Dd % 32 || n_seg > 65535) return (int)hipErrorInvalidValue;
    const int tiles = ((max_rows + 63) / 64) * ((max_rows + 15) / 16);
    hipLaunchKernelGGL(pair_scores_blockdiag_kernel, dim3((unsigned)tiles, (unsigned)n_seg), dim3(256), 0, (hipStream_t)stream, x, seg,
                       out_off, w, bias, out, Dd);
    return (int)hipGetLastError();
}

int seam_match_scores_f32(const float* logits, float* score, int64_t n_pairs, void* stream) {
    if (n_pairs <= 0) return 0;
    int grid = (int)((n_pairs + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(match_scores_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, score, (size_t)n_pairs);
    return (int)hipGetLastError();
}

int seam_rank_of_f32(const float* logits, const int64_t* target, int64_t* rank, int Q, int G, void* stream) {
    if (Q <= 0) return 0;
    hipLaunchKernelGGL(rank_of_kernel, dim3(Q), dim3(256), 0, (hipStream_t)stream, logits, target, rank, G);
    return (int)hipGetLastError();
}

int seam_score_reduce_f32(const float* score, float* out, int n, int G, int mode, void* stream) {
    if (G <= 0) return 0;
    if (n <= 0 || mode < 0 || mode > 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(score_reduce_kernel, dim3((G + 255) / 256), dim3(256), 0, (hipStream_t)stream, score, out, n, G, mode);
    return (int)hipGetLastError();
}

int seam_score_reduce_seg_f32(const float* score, const int* seg, float* out, int P, int G, int mode, void* stream) {
    if (G <= 0 || P <= 0) return 0;
    if (mode < 0 || mode > 1 || P > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(score_reduce_seg_kernel, dim3((G + 255) / 256, P), dim3(256), 0, (hipStream_t)stream, score, seg, out, G, mode);
    return (int)hipGetLastError();
}

int seam_rank_of_scores_f32(const float* score, const int64_t* target, int64_t* rank, int Q, int G, void* stream) {
    if (Q <= 0) return 0;
    hipLaunchKernelGGL(rank_of_scores_kernel, dim3(Q), dim3(256), 0, (hipStream_t)stream, score, target, rank, G);
    return (int)hipGetLastError();
}

int64_t seam_pair_topk_workspace_floats(int Q, int G, int k) {
    const int64_t nseg = (G + TK_SEG - 1) / TK_SEG;
    return (int64_t)Q * nseg * k * 3 + 16;
}

int seam_pair_topk_f32(const float* a, const float* b, const float* w, const float* bias, int64_t* idx, float* score,
                       int Q, int G, int Dd, int k, float* ws, void* stream) {
    if (Q <= 0 || k <= 0) return 0;
    if (k > G || k > TK_SEG || (Dd % 32)) return (int)hipErrorInvalidValue;
    const int nseg = (G + TK_SEG - 1) / TK_SEG;
    hipLaunchKernelGGL(pair_topk_stage1, dim3(nseg, (Q + 31) / 32), dim3(256), 0, (hipStream_t)stream, a, b, w, bias, ws,
                       Q, G, Dd, k, nseg);
    hipLaunchKernelGGL(pair_topk_stage2, dim3(Q), dim3(256), 0, (hipStream_t)stream, ws, idx, score, nseg * k, k);
    return (int)hipGetLastError();
}

int seam_rank_topk_f32(const float* logits, int64_t* idx, float* score, int Q, int G, int k, void* stream) {
    if (Q <= 0 || k <= 0) return 0;
    if (k > G) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rank_topk_kernel, dim3(Q), dim3(256), 0, (hipStream_t)stream, logits, idx, score, G, k);
    return (int)hipGetLastError();
}

}  // extern "C"
