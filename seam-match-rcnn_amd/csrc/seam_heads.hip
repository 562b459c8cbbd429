// seam_heads.hip -- SEAM temporal head kernels for gfx950 (wave64):
//   nlb_attnpool  concatenation-form non-local block + softmax attention pooling, one workgroup
//                 per sequence, all intermediates on chip (reference: ~12 launches + a
//                 [1,256,T,T] temporary PER SEQUENCE inside a Python loop)
//   pair_logits   x5[i,j,:] = W * (a_i - b_j)^2 + bias, register-blocked direct form (the direct
//                 form is the parity reference; no a^2+b^2-2ab cancellation)
//   rank_topk     descending rank of softmax(x5)[...,1], k rounds of a workgroup arg-max
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

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
    const int T = p.len[s];
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
// pair logits: block tile (8*QT) x (32*GT) pairs, thread tile QT x GT, k chunks of 32 through LDS
template <int QT, int GT>
__global__ __launch_bounds__(256) void pair_logits_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int Q, int G, int Dd) {
    constexpr int KC = 32, LD = KC + 4;
    constexpr int BQ = 8 * QT, BG = 32 * GT;
    __shared__ __attribute__((aligned(16))) float as[BQ * LD];
    __shared__ __attribute__((aligned(16))) float bs[BG * LD];
    __shared__ __attribute__((aligned(16))) float ws[2 * KC];
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;
    const int q0 = blockIdx.y * BQ, g0 = blockIdx.x * BG;
    float acc0[QT][GT], acc1[QT][GT];
#pragma unroll
    for (int i = 0; i < QT; ++i)
#pragma unroll
        for (int j = 0; j < GT; ++j) { acc0[i][j] = 0.f; acc1[i][j] = 0.f; }

    for (int k0 = 0; k0 < Dd; k0 += KC) {
        __syncthreads();
        for (int i = tid; i < BQ * (KC / 4); i += 256) {
            const int r = i / (KC / 4), c = i % (KC / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q0 + r < Q) v = *reinterpret_cast<const f32x4*>(a + (size_t)(q0 + r) * Dd + k0 + c * 4);
            *reinterpret_cast<f32x4*>(&as[r * LD + c * 4]) = v;
        }
        for (int i = tid; i < BG * (KC / 4); i += 256) {
            const int r = i / (KC / 4), c = i % (KC / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g0 + r < G) v = *reinterpret_cast<const f32x4*>(b + (size_t)(g0 + r) * Dd + k0 + c * 4);
            *reinterpret_cast<f32x4*>(&bs[r * LD + c * 4]) = v;
        }
        if (tid < 2 * KC) ws[tid] = w[(size_t)(tid / KC) * Dd + k0 + (tid % KC)];
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < KC; k4 += 4) {
            f32x4 av[QT], bv[GT];
#pragma unroll
            for (int i = 0; i < QT; ++i) av[i] = *reinterpret_cast<const f32x4*>(&as[(ty * QT + i) * LD + k4]);
#pragma unroll
            for (int j = 0; j < GT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(&bs[(tx + 32 * j) * LD + k4]);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(&ws[k4]);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(&ws[KC + k4]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < QT; ++i)
#pragma unroll
                    for (int j = 0; j < GT; ++j) {
                        const float d = av[i][kk] - bv[j][kk];
                        const float d2 = d * d;
                        acc0[i][j] = fmaf(d2, w0[kk], acc0[i][j]);
                        acc1[i][j] = fmaf(d2, w1[kk], acc1[i][j]);
                    }
        }
    }
    const float b0 = bias[0], b1 = bias[1];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const int qi = q0 + ty * QT + i;
        if (qi >= Q) continue;
#pragma unroll
        for (int j = 0; j < GT; ++j) {
            const int gj = g0 + tx + 32 * j;
            if (gj < G) {
                float2 v = make_float2(acc0[i][j] + b0, acc1[i][j] + b1);
                *reinterpret_cast<float2*>(out + ((size_t)qi * G + gj) * 2) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rank_topk: one workgroup per query; k rounds of arg-max over d = x1 - x0 with the strict total
// order (d desc, index asc); the previous winner bounds the next round (no scratch, no mutation).
__global__ __launch_bounds__(256) void rank_topk_kernel(const float* __restrict__ logits, int64_t* __restrict__ idx,
                                                        float* __restrict__ score, int G, int k) {
    __shared__ float rv[4];
    __shared__ int ri[4];
    __shared__ float bestv;
    __shared__ int besti;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float2* row = reinterpret_cast<const float2*>(logits) + (size_t)q * G;
    float pv = INFINITY;
    int pi = -1;
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int g = tid; g < G; g += 256) {
            const float2 x = row[g];
            float d = x.y - x.x;
            if (d != d) d = -INFINITY;                                  // NaN ranks last
            const bool elig = (d < pv) || (d == pv && g > pi);           // strictly after the previous winner
            if (elig && (d > bv || (d == bv && g < bi))) { bv = d; bi = g; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { rv[wid] = bv; ri[wid] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = rv[0];
            int i = ri[0];
            for (int w = 1; w < 4; ++w)
                if (rv[w] > v || (rv[w] == v && ri[w] < i)) { v = rv[w]; i = ri[w]; }
            bestv = v;
            besti = i;
            const bool found = i != 0x7fffffff;
            idx[(size_t)q * k + round] = found ? (int64_t)i : (int64_t)-1;
            float sc = 0.f;
            if (found) {
                const float2 x = row[i];
                const float mx = fmaxf(x.x, x.y);                        // softmax(x)[1]
                const float e0 = expf(x.x - mx), e1 = expf(x.y - mx);
                sc = e1 / (e0 + e1);
            }
            score[(size_t)q * k + round] = sc;
        }
        __syncthreads();
        pv = bestv;
        pi = besti;
        __syncthreads();
    }
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
    constexpr int KC = 32, LD = KC + 4, BQ = 32, BG = 128;
    __shared__ __attribute__((aligned(16))) float as[BQ * LD];
    __shared__ __attribute__((aligned(16))) float bs[BG * LD];
    __shared__ __attribute__((aligned(16))) float ws[2 * KC];
    __shared__ float L0[BQ][TK_SEG + 1];
    __shared__ float L1[BQ][TK_SEG + 1];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5, lane = tid & 63, wid = tid >> 6;
    const int seg = blockIdx.x, q0 = blockIdx.y * BQ, gseg = seg * TK_SEG;
    const float b0 = bias[0], b1 = bias[1];
    for (int sub = 0; sub < TK_SEG / BG; ++sub) {
        const int g0 = gseg + sub * BG;
        float acc0[4][4], acc1[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc0[i][j] = 0.f; acc1[i][j] = 0.f; }
        for (int k0 = 0; k0 < Dd; k0 += KC) {
            __syncthreads();
            for (int i = tid; i < BQ * (KC / 4); i += 256) {
                const int r = i / (KC / 4), c = i % (KC / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q0 + r < Q) v = *reinterpret_cast<const f32x4*>(a + (size_t)(q0 + r) * Dd + k0 + c * 4);
                *reinterpret_cast<f32x4*>(&as[r * LD + c * 4]) = v;
            }
            for (int i = tid; i < BG * (KC / 4); i += 256) {
                const int r = i / (KC / 4), c = i % (KC / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (g0 + r < G) v = *reinterpret_cast<const f32x4*>(b + (size_t)(g0 + r) * Dd + k0 + c * 4);
                *reinterpret_cast<f32x4*>(&bs[r * LD + c * 4]) = v;
            }
            if (tid < 2 * KC) ws[tid] = w[(size_t)(tid / KC) * Dd + k0 + (tid % KC)];
            __syncthreads();
#pragma unroll
            for (int k4 = 0; k4 < KC; k4 += 4) {
                f32x4 av[4], bv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const f32x4*>(&as[(ty * 4 + i) * LD + k4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(&bs[(tx + 32 * j) * LD + k4]);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(&ws[k4]);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(&ws[KC + k4]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float d = av[i][kk] - bv[j][kk];
                            const float d2 = d * d;
                            acc0[i][j] = fmaf(d2, w0[kk], acc0[i][j]);
                            acc1[i][j] = fmaf(d2, w1[kk], acc1[i][j]);
                        }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                L0[ty * 4 + i][sub * BG + tx + 32 * j] = acc0[i][j] + b0;
                L1[ty * 4 + i][sub * BG + tx + 32 * j] = acc1[i][j] + b1;
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
    __shared__ float rv[4];
    __shared__ int ri[4], rc[4];
    __shared__ float bestv;
    __shared__ int besti, bestc;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* c = cand + (size_t)q * ncand * 3;
    float pv = INFINITY;
    int pi = -1;
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff, bc = -1;
        for (int j = tid; j < ncand; j += 256) {
            const int g = __float_as_int(c[j * 3 + 2]);
            if (g < 0) continue;
            float d = c[j * 3 + 1] - c[j * 3];
            if (d != d) d = -INFINITY;
            const bool elig = (d < pv) || (d == pv && g > pi);
            if (elig && tk_better(d, g, bv, bi)) { bv = d; bi = g; bc = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            const int oc = __shfl_xor(bc, o, 64);
            if (tk_better(ov, oi, bv, bi)) { bv = ov; bi = oi; bc = oc; }
        }
        if (lane == 0) { rv[wid] = bv; ri[wid] = bi; rc[wid] = bc; }
        __syncthreads();
        if (tid == 0) {
            float v = rv[0];
            int i = ri[0], cc = rc[0];
            for (int w2 = 1; w2 < 4; ++w2)
                if (tk_better(rv[w2], ri[w2], v, i)) { v = rv[w2]; i = ri[w2]; cc = rc[w2]; }
            bestv = v; besti = i; bestc = cc;
            const bool found = i != 0x7fffffff;
            idx[(size_t)q * k + round] = found ? (int64_t)i : (int64_t)-1;
            float sc = 0.f;
            if (found) {
                const float x0 = c[cc * 3], x1 = c[cc * 3 + 1];
                const float mx = fmaxf(x0, x1);
                const float e0 = expf(x0 - mx), e1 = expf(x1 - mx);
                sc = e1 / (e0 + e1);
            }
            score[(size_t)q * k + round] = sc;
        }
        __syncthreads();
        pv = bestv;
        pi = besti;
        __syncthreads();
    }
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
