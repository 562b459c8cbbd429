// seam_backward.hip -- gradient kernels of the SEAM match heads: what the training
// caller (ref stuffs/engine.py:120-121,158-168,183-185) runs through MatchPredictor / TemporalAggregationNLB
// in .train() mode.  fp32 throughout, every reduction in a fixed order (no atomics): bit-reproducible.
//
//   conv / linear   input grad  = the forward implicit-GEMM kernel on 180-degree rotated, channel-swapped
//                                 weights (seam_pack_conv_weight_f32 mode 2) with the ReLU mask in its epilogue
//                   weight grad = conv_wgrad_kernel below: fp32-MFMA GEMM  dW[k, (r,s), c] = sum_pix dY[pix,k] *
//                                 X[pix+(r,s), c], split over the pixel axis, partials reduced by a second kernel
//                   bias grad   = colsum_kernel
//   avg-pool+ReLU, BatchNorm1d (batch statistics), pairwise classifier, non-local block + attention pooling:
//   small fused kernels (training batches are a few dozen ROIs / a few sequences).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr unsigned kOob = 0x80000000u;
constexpr int D = 256;
constexpr int DI = 128;
constexpr int TB = 64;          // longest sequence the NLB backward keeps in LDS

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of one value per thread (256 threads), result broadcast; `red` = 4 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
    const float* x;      // NHWC [N,H,W,C]
    const float* dy;     // [M = N*Ho*Wo, K]
    float* ws;           // [splits][taps][K][C]
    int N, H, W, C, Ho, Wo, K, R, S, stride, pad;
    int M, tiles_k, tiles_c, nchunks, chunks_per_split;
};

// One block: a 128 (k) x 128 (c) tile of one tap (r,s), reduced over this split's pixels in chunks of 32.
// LDS tiles are [pixel][channel]: an MFMA operand lane reads ONE float (row = channel lane&31, k-slot =
// pixel parity lane>>5) -- consecutive lanes, consecutive banks -- so no transpose of dY or X is ever made.
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs p) {
    __shared__ __attribute__((aligned(16))) float As[2][32 * 128];
    __shared__ __attribute__((aligned(16))) float Bs[2][32 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    int b = blockIdx.x;
    const int tc = b % p.tiles_c; b /= p.tiles_c;
    const int tk = b % p.tiles_k;
    const int tap = b / p.tiles_k;
    const int r = tap / p.S, s = tap - r * p.S;
    const int k0 = tk * 128, c0 = tc * 128;
    const int ch0 = blockIdx.y * p.chunks_per_split;
    const int ch1 = min(p.nchunks, ch0 + p.chunks_per_split);

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.M * p.K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((unsigned)p.N * p.H * p.W * p.C * 4u), 0x00020000);
    const int col = tid & 31, rowb = tid >> 5;
    const bool kin = k0 + col * 4 < p.K, cin = c0 + col * 4 < p.C;
    const int HoWo = p.Ho * p.Wo;

    f32x4 ar[4], br[4];
    auto load = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix = ch * 32 + rowb + 8 * i;
            const bool in = pix < p.M && ch < ch1;
            const unsigned oa = (in && kin) ? (unsigned)((pix * p.K + k0 + col * 4) * 4) : kOob;
            ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, oa, 0, 0));
            const int n = pix / HoWo, rm = pix - n * HoWo;
            const int ho = rm / p.Wo, wo = rm - ho * p.Wo;
            const int hi = ho * p.stride + r - p.pad, wi = wo * p.stride + s - p.pad;
            const bool ok = in && cin && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned ob = ok ? (unsigned)((((n * p.H + hi) * p.W + wi) * p.C + c0 + col * 4) * 4) : kOob;
            br[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, ob, 0, 0));
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&As[buf][(rowb + 8 * i) * 128 + col * 4]) = ar[i];
            *reinterpret_cast<f32x4*>(&Bs[buf][(rowb + 8 * i) * 128 + col * 4]) = br[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    load(ch0);
    store(0);
    __syncthreads();
    const int fo = (lane >> 5) * 128 + (lane & 31);
    for (int ch = ch0; ch < ch1; ++ch) {
        const int buf = (ch - ch0) & 1;
        load(ch + 1);                       // unconditional: past the split's end it returns zeros nobody reads
        const float* a = &As[buf][fo + wm0];
        const float* bb = &Bs[buf][fo + wn0];
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            const float a0 = a[pp * 256], a1 = a[pp * 256 + 32];
            const float b0 = bb[pp * 256], b1 = bb[pp * 256 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        store(buf ^ 1);
        __syncthreads();
    }

    const int taps = p.R * p.S;
    float* out = p.ws + ((size_t)blockIdx.y * taps + tap) * p.K * p.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wn0 + j * 32 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = k0 + wm0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (k < p.K && c < p.C) out[(size_t)k * p.C + c] = acc[i][j][q];
            }
        }
}

// dw[k][c][tap] (PyTorch OIHW) = sum over splits of ws[split][tap][k][c]
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int taps, int K, int C) {
    const size_t per = (size_t)taps * K * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t rest = i / C;
        const int k = (int)(rest % K);
        const int tap = (int)(rest / K);
        float acc = 0.f;
        for (int sidx = 0; sidx < splits; ++sidx) acc += ws[(size_t)sidx * per + i];
        dw[((size_t)k * C + c) * taps + tap] = acc;
    }
}

// out[k] = sum_m x[m][k]   (bias gradients), two fixed-order stages: blockIdx.y sums a contiguous slab of rows into
// part[y][k] (256 threads = 4 row groups x 64 columns), then colsum_final adds the slabs.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ partial, int M, int K,
                                                     int rows_per) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float acc = 0.f;
    if (c < K)
        for (int m = m0 + g; m < m1; m += 4) acc += x[(size_t)m * K + c];
    part[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && c < K)
        partial[(size_t)blockIdx.y * K + c] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

__global__ void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int slabs, int K) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K) return;
    float acc = 0.f;
    for (int i = 0; i < slabs; ++i) acc += partial[(size_t)i * K + c];
    out[c] = acc;
}

// ------------------------------------------------------------------------------------------------ pool / BN
// AvgPool2d over the whole HW map followed by ReLU, and the ReLU in front of it (ref models/match_head.py:57-60):
// dy[n][hw][c] = y[n][hw][c] > 0 ? dpool[n][c] / HW : 0     (pool > 0 whenever any y > 0, so its ReLU mask is implied)
__global__ void avgpool_relu_bwd_kernel(const float* __restrict__ dpool, const float* __restrict__ y, float* __restrict__ dy,
                                        int HW, int C, size_t total) {
    const float inv = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t n = i / ((size_t)HW * C);
        dy[i] = y[i] > 0.f ? dpool[n * C + c] * inv : 0.f;
    }
}

// BatchNorm1d, training mode (nn.BatchNorm1d(256), ref models/match_head.py:62): batch statistics (biased
// variance for the normalisation, unbiased for running_var), momentum update of the running buffers.
__global__ void bn1d_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      float* __restrict__ y, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                      float* __restrict__ run_mean, float* __restrict__ run_var, int M, int F, float momentum,
                                      float eps) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float mean = 0.f;
    for (int m = 0; m < M; ++m) mean += x[(size_t)m * F + f];
    mean /= (float)M;
    float var = 0.f;
    for (int m = 0; m < M; ++m) {
        const float d = x[(size_t)m * F + f] - mean;
        var = fmaf(d, d, var);
    }
    const float vb = var / (float)M;
    const float inv = 1.f / sqrtf(vb + eps);
    const float g = gamma[f], bta = beta[f];
    for (int m = 0; m < M; ++m) y[(size_t)m * F + f] = (x[(size_t)m * F + f] - mean) * inv * g + bta;
    save_mean[f] = mean;
    save_invstd[f] = inv;
    if (run_mean) {
        run_mean[f] = (1.f - momentum) * run_mean[f] + momentum * mean;
        run_var[f] = (1.f - momentum) * run_var[f] + momentum * (var / (float)(M - 1));
    }
}

__global__ void bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ save_mean,
                                const float* __restrict__ save_invstd, const float* __restrict__ gamma, float* __restrict__ dx,
                                float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int F, int frozen) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float mean = save_mean[f], inv = save_invstd[f];
    float sg = 0.f, sb = 0.f;
    for (int m = 0; m < M; ++m) {
        const float g = dy[(size_t)m * F + f];
        sb += g;
        sg = fmaf(g, (x[(size_t)m * F + f] - mean) * inv, sg);
    }
    dgamma[f] = sg;
    dbeta[f] = sb;
    if (frozen) {          // eval-mode statistics (mean / invstd are constants): a plain affine map
        const float k0 = gamma[f] * inv;
        for (int m = 0; m < M; ++m) dx[(size_t)m * F + f] = k0 * dy[(size_t)m * F + f];
        return;
    }
    const float k = gamma[f] * inv / (float)M;
    for (int m = 0; m < M; ++m) {
        const float xh = (x[(size_t)m * F + f] - mean) * inv;
        dx[(size_t)m * F + f] = k * ((float)M * dy[(size_t)m * F + f] - sb - xh * sg);
    }
}

// ------------------------------------------------------------------------------------------------ pair logits
// x5[i,j,c] = sum_d w[c,d] (a[i,d]-b[j,d])^2 + bias[c]  (ref models/match_head.py:73-74,161-162)
// da[i,d] =  2 sum_j (a-b) (g0 w0d + g1 w1d);   db[j,d] = -2 sum_i (...)
__global__ __launch_bounds__(256) void pair_bwd_ab_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ w, const float* __restrict__ g,
                                                          float* __restrict__ da, float* __restrict__ db, int Q, int G) {
    const int d = threadIdx.x;
    const float w0 = w[d], w1 = w[D + d];
    const int row = blockIdx.x;
    if (row < Q) {
        const float ai = a[(size_t)row * D + d];
        float acc = 0.f;
        for (int j = 0; j < G; ++j) {
            const float2 gg = reinterpret_cast<const float2*>(g)[(size_t)row * G + j];
            acc = fmaf(ai - b[(size_t)j * D + d], gg.x * w0 + gg.y * w1, acc);
        }
        da[(size_t)row * D + d] = 2.f * acc;
    } else {
        const int j = row - Q;
        const float bj = b[(size_t)j * D + d];
        float acc = 0.f;
        for (int i = 0; i < Q; ++i) {
            const float2 gg = reinterpret_cast<const float2*>(g)[(size_t)i * G + j];
            acc = fmaf(a[(size_t)i * D + d] - bj, gg.x * w0 + gg.y * w1, acc);
        }
        db[(size_t)j * D + d] = -2.f * acc;
    }
}

// dw[c,d] = sum_ij g[i,j,c] (a-b)^2 ; block = one d, threads stride the pairs;  block D: dbias[c] = sum_ij g[i,j,c]
__global__ __launch_bounds__(256) void pair_bwd_w_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ g, float* __restrict__ dw,
                                                         float* __restrict__ dbias, int Q, int G) {
    __shared__ float red[4];
    const int d = blockIdx.x;
    const long n = (long)Q * G;
    float s0 = 0.f, s1 = 0.f;
    for (long pidx = threadIdx.x; pidx < n; pidx += 256) {
        const float2 gg = reinterpret_cast<const float2*>(g)[pidx];
        float sq = 1.f;
        if (d < D) {
            const int i = (int)(pidx / G), j = (int)(pidx - (long)i * G);
            const float df = a[(size_t)i * D + d] - b[(size_t)j * D + d];
            sq = df * df;
        }
        s0 = fmaf(gg.x, sq, s0);
        s1 = fmaf(gg.y, sq, s1);
    }
    s0 = block_sum(s0, red);
    s1 = block_sum(s1, red);
    if (threadIdx.x == 0) {
        if (d < D) { dw[d] = s0; dw[D + d] = s1; }
        else { dbias[0] = s0; dbias[1] = s1; }
    }
}

// ------------------------------------------------------------------------------------------------ NLB + attention pooling
struct NlbBwdArgs {
    const float* seq;          // X rows: seq + s*s_stride + t*t_stride
    int64_t t_stride, s_stride;
    const int* len;
    int S, Tmax;
    const float* w_proj_t;     // [256][384]  (theta | phi | g)^T
    const float* b_proj;       // [384]
    const float* w_cat;        // [256]
    const float* w_out_t;      // [128][256]
    const float* b_out;        // [256]
    const float* w_att;        // [256]
    const float* b_att;        // [1]
    const float* dout;         // [S][256]
    const float* dz_ext;       // optional [row = s*dz_s_stride + t*dz_t_stride][256]: the gradient w.r.t. the block's OUTPUT rows z
    int64_t dz_t_stride, dz_s_stride;   //   (direct callers of NONLocalBlock1D.forward); replaces the pooling backward, dout unused
    float* dseq;               // same strides as seq
    // per-row scratch, row id = s*Tmax + t
    float* G;                  // [rows][128]
    float* Y;                  // [rows][128]
    float* Z;                  // [rows][256]
    float* dZn;                // [rows][256]  (zero for NLB-bypassed rows)
    float* dY;                 // [rows][128]
    float* dG;                 // [rows][128]
    float* vec;                // [rows][3]: da, db, de
    int use_nlb;
};

// One block per sequence: recompute the forward (ref models/nlb.py:66-101 closed form, models/match_head.py:119-121),
// then back-propagate dout[s] to the sequence rows; per-row factors of the parameter gradients go to scratch.
__global__ __launch_bounds__(256) void nlb_bwd_seq_kernel(const NlbBwdArgs p) {
    __shared__ float xs[D];
    __shared__ float red[4];
    __shared__ float av[TB], bv[TB], ev[TB], sv[TB], dav[TB], dbv[TB];
    __shared__ float dS[TB][TB + 1];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int T = min(p.len[s], p.Tmax);              // a length beyond the packed rows would index LDS / scratch out of range
    if (T <= 0) return;
    const float* X = p.seq + (int64_t)s * p.s_stride;
    float* dX = p.dseq + (int64_t)s * p.s_stride;
    const size_t r0 = (size_t)s * p.Tmax;
    const bool nlb = p.use_nlb == 2 || (p.use_nlb && T > 1);
    const float Tf = (float)T;
    const bool pooled = p.dz_ext == nullptr;
    const float wa = pooled ? p.w_att[tid] : 0.f, ba = pooled ? p.b_att[0] : 0.f;

    if (nlb) {
        const float wc = p.w_cat[tid], bp0 = p.b_proj[tid], bp1 = tid < DI ? p.b_proj[256 + tid] : 0.f;
        for (int t = 0; t < T; ++t) {
            __syncthreads();
            xs[tid] = X[(int64_t)t * p.t_stride + tid];
            __syncthreads();
            float a0 = 0.f, a1 = 0.f;
            for (int k = 0; k < D; ++k) {
                const float xv = xs[k];
                a0 = fmaf(xv, p.w_proj_t[(size_t)k * 384 + tid], a0);
                if (tid < DI) a1 = fmaf(xv, p.w_proj_t[(size_t)k * 384 + 256 + tid], a1);
            }
            const float v = wave_sum((a0 + bp0) * wc);
            __syncthreads();
            if (lane == 0) red[wid] = v;
            if (tid < DI) p.G[(r0 + t) * DI + tid] = a1 + bp1;
            __syncthreads();
            if (tid == 0) { av[t] = red[0] + red[1]; bv[t] = red[2] + red[3]; }
        }
        __syncthreads();
        // Y = f G
        {
            const int c = tid & (DI - 1);
            for (int i = tid >> 7; i < T; i += 2) {
                float y = 0.f;
                for (int j = 0; j < T; ++j) y = fmaf(fmaxf(av[i] + bv[j], 0.f) / Tf, p.G[(r0 + j) * DI + c], y);
                p.Y[(r0 + i) * DI + c] = y;
            }
        }
        __syncthreads();
    }
    // Z rows + attention scores
    for (int t = 0; t < T; ++t) {
        float z = X[(int64_t)t * p.t_stride + tid];
        if (nlb) {
            float acc = 0.f;
            for (int c = 0; c < DI; ++c) acc = fmaf(p.Y[(r0 + t) * DI + c], p.w_out_t[(size_t)c * D + tid], acc);
            z += acc + p.b_out[tid];
        }
        p.Z[(r0 + t) * D + tid] = z;
        const float e = block_sum(z * wa, red);
        if (tid == 0) ev[t] = e + ba;
    }
    __syncthreads();
    // softmax over t; ds_t = dout . Z_t ; de = s (ds - sum s ds)
    const float dov = pooled ? p.dout[(size_t)s * D + tid] : 0.f;
    if (pooled) {
        float m = -INFINITY;
        for (int t = 0; t < T; ++t) m = fmaxf(m, ev[t]);
        float l = 0.f;
        for (int t = 0; t < T; ++t) l += expf(ev[t] - m);
        for (int t = 0; t < T; ++t) {
            const float ds = block_sum(dov * p.Z[(r0 + t) * D + tid], red);
            if (tid == 0) { sv[t] = expf(ev[t] - m) / l; dav[t] = ds; }  // dav: temporary home of ds
        }
        __syncthreads();
        float dot = 0.f;
        for (int t = 0; t < T; ++t) dot = fmaf(sv[t], dav[t], dot);
        __syncthreads();
        if (tid < T) {
            const float de = sv[tid] * (dav[tid] - dot);
            dbv[tid] = de;                                               // dbv: temporary home of de
            p.vec[(r0 + tid) * 3 + 2] = de;
        }
    } else if (tid < T) {
        p.vec[(r0 + tid) * 3 + 2] = 0.f;                                 // no attention scorer behind a direct block call
    }
    __syncthreads();
    // dZ_t = s_t dout + de_t wa   (or handed in by the caller)
    for (int t = 0; t < T; ++t) {
        const float dz = pooled ? fmaf(sv[t], dov, dbv[t] * wa)
                                : p.dz_ext[(int64_t)s * p.dz_s_stride + (int64_t)t * p.dz_t_stride + tid];
        dX[(int64_t)t * p.t_stride + tid] = dz;                          // residual path (the whole gradient when bypassed)
        p.dZn[(r0 + t) * D + tid] = nlb ? dz : 0.f;
    }
    if (!nlb) {
        if (tid < T) { p.vec[(r0 + tid) * 3] = 0.f; p.vec[(r0 + tid) * 3 + 1] = 0.f; }
        for (int t = 0; t < T; ++t)
            if (tid < DI) { p.dG[(r0 + t) * DI + tid] = 0.f; p.Y[(r0 + t) * DI + tid] = 0.f; }
        return;
    }
    __syncthreads();
    // dY_t[c] = sum_d dZ_t[d] Ww[d][c]
    for (int t = 0; t < T; ++t) {
        __syncthreads();
        xs[tid] = p.dZn[(r0 + t) * D + tid];
        __syncthreads();
        if (tid < DI) {
            float acc = 0.f;
            for (int d = 0; d < D; ++d) acc = fmaf(xs[d], p.w_out_t[(size_t)tid * D + d], acc);
            p.dY[(r0 + t) * DI + tid] = acc;
        }
    }
    __syncthreads();
    // dS[i][j] = (dY_i . G_j) [a_i + b_j > 0] / T
    for (int ij = tid; ij < T * T; ij += 256) {
        const int i = ij / T, j = ij - i * T;
        float acc = 0.f;
        if (av[i] + bv[j] > 0.f) {
            for (int c = 0; c < DI; ++c) acc = fmaf(p.dY[(r0 + i) * DI + c], p.G[(r0 + j) * DI + c], acc);
            acc /= Tf;
        }
        dS[i][j] = acc;
    }
    __syncthreads();
    if (tid < T) {
        float sa = 0.f, sb = 0.f;
        for (int j = 0; j < T; ++j) { sa += dS[tid][j]; sb += dS[j][tid]; }
        dav[tid] = sa;
        dbv[tid] = sb;
        p.vec[(r0 + tid) * 3] = sa;
        p.vec[(r0 + tid) * 3 + 1] = sb;
    }
    // dG_j[c] = sum_i f[i][j] dY_i[c]
    {
        const int c = tid & (DI - 1);
        for (int j = tid >> 7; j < T; j += 2) {
            float acc = 0.f;
            for (int i = 0; i < T; ++i) acc = fmaf(fmaxf(av[i] + bv[j], 0.f) / Tf, p.dY[(r0 + i) * DI + c], acc);
            p.dG[(r0 + j) * DI + c] = acc;
        }
    }
    __syncthreads();
    // dX_t += da_t u_a + db_t u_b + dG_t Wg,   u_a = Wth^T wc[:128], u_b = Wph^T wc[128:]
    float ua = 0.f, ub = 0.f;
    for (int c = 0; c < DI; ++c) {
        ua = fmaf(p.w_cat[c], p.w_proj_t[(size_t)tid * 384 + c], ua);
        ub = fmaf(p.w_cat[DI + c], p.w_proj_t[(size_t)tid * 384 + DI + c], ub);
    }
    for (int t = 0; t < T; ++t) {
        float acc = fmaf(dav[t], ua, dbv[t] * ub);
        for (int c = 0; c < DI; ++c) acc = fmaf(p.dG[(r0 + t) * DI + c], p.w_proj_t[(size_t)tid * 384 + 256 + c], acc);
        dX[(int64_t)t * p.t_stride + tid] += acc;
    }
}

struct NlbParamArgs {
    NlbBwdArgs b;
    // outputs in the reference's parameter layouts
    float* d_theta_w; float* d_theta_b; float* d_phi_w; float* d_phi_b; float* d_g_w; float* d_g_b;   // [128][256], [128]
    float* d_cat;              // [256]
    float* d_W_w; float* d_W_b;   // [256][128], [256]
    float* d_att_w; float* d_att_b;   // [256], [1]
    float* tmp;                // [2*256 + 2]: va, vb, sa, sb
};

__device__ __forceinline__ bool row_live(const NlbBwdArgs& b, int row) { return (row % b.Tmax) < b.len[row / b.Tmax]; }
__device__ __forceinline__ const float* x_row(const NlbBwdArgs& b, int row) {
    return b.seq + (int64_t)(row / b.Tmax) * b.s_stride + (int64_t)(row % b.Tmax) * b.t_stride;
}

// grid: 128 blocks (dWg rows c) + 256 blocks (dWw rows d) + 1 block (vector sums); rows summed in index order.
__global__ __launch_bounds__(256) void nlb_param_grad_kernel(const NlbParamArgs p) {
    const NlbBwdArgs& b = p.b;
    const int rows = b.S * b.Tmax, tid = threadIdx.x, blk = blockIdx.x;
    if (blk < DI) {                                   // dWg[c][d] = sum dG[row][c] X[row][d]
        float acc = 0.f;
        for (int row = 0; row < rows; ++row)
            if (row_live(b, row)) acc = fmaf(b.dG[(size_t)row * DI + blk], x_row(b, row)[tid], acc);
        p.d_g_w[(size_t)blk * D + tid] = acc;
    } else if (blk < DI + D) {                        // dWw[d][c] = sum dZn[row][d] Y[row][c]
        const int d = blk - DI;
        if (tid < DI) {
            float acc = 0.f;
            for (int row = 0; row < rows; ++row)
                if (row_live(b, row)) acc = fmaf(b.dZn[(size_t)row * D + d], b.Y[(size_t)row * DI + tid], acc);
            p.d_W_w[(size_t)d * DI + tid] = acc;
        }
    } else {
        float va = 0.f, vb = 0.f, dwa = 0.f, dbw = 0.f, dbg = 0.f, sa = 0.f, sb = 0.f, se = 0.f;
        for (int row = 0; row < rows; ++row) {
            if (!row_live(b, row)) continue;
            const float da = b.vec[(size_t)row * 3], db = b.vec[(size_t)row * 3 + 1], de = b.vec[(size_t)row * 3 + 2];
            const float xv = x_row(b, row)[tid];
            va = fmaf(da, xv, va);
            vb = fmaf(db, xv, vb);
            dwa = fmaf(de, b.Z[(size_t)row * D + tid], dwa);
            dbw += b.dZn[(size_t)row * D + tid];
            if (tid < DI) dbg += b.dG[(size_t)row * DI + tid];
            sa += da; sb += db; se += de;
        }
        p.tmp[tid] = va;
        p.tmp[D + tid] = vb;
        p.d_att_w[tid] = dwa;
        p.d_W_b[tid] = dbw;
        if (tid < DI) p.d_g_b[tid] = dbg;
        if (tid == 0) { p.tmp[2 * D] = sa; p.tmp[2 * D + 1] = sb; p.d_att_b[0] = se; }
    }
}

// theta/phi enter only through a = TH.wc1, b = PH.wc2:  dW = wc (x) v,  db = s wc,  dwc = W v + bias s
__global__ __launch_bounds__(256) void nlb_param_assemble_kernel(const NlbParamArgs p) {
    const NlbBwdArgs& b = p.b;
    const int c = blockIdx.x & (DI - 1), which = blockIdx.x >> 7, tid = threadIdx.x;      // which: 0 theta, 1 phi
    __shared__ float red[4];
    const float* v = p.tmp + which * D;
    const float sc = p.tmp[2 * D + which];
    const float wc = b.w_cat[which * DI + c];
    float* dw = which ? p.d_phi_w : p.d_theta_w;
    dw[(size_t)c * D + tid] = wc * v[tid];
    const float dot = block_sum(b.w_proj_t[(size_t)tid * 384 + which * DI + c] * v[tid], red);
    if (tid == 0) {
        (which ? p.d_phi_b : p.d_theta_b)[c] = sc * wc;
        p.d_cat[which * DI + c] = dot + b.b_proj[which * DI + c] * sc;
    }
}

// nn.CrossEntropyLoss(weight=w) on [n,2] logits, mean reduction: loss = sum_i w[y_i] (lse_i - x_i[y_i]) / sum_i w[y_i];
// dlogits_i = w[y_i] / sum_w * (softmax(x_i) - onehot(y_i)).  One block; fixed reduction order.
__global__ __launch_bounds__(256) void ce2_kernel(const float* __restrict__ x, const int64_t* __restrict__ y,
                                                  const float* __restrict__ w, float* __restrict__ loss,
                                                  float* __restrict__ dx, long n) {
    __shared__ float red[4];
    const float w0 = w[0], w1 = w[1];
    float sl = 0.f, sw = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float2 v = reinterpret_cast<const float2*>(x)[i];
        const float m = fmaxf(v.x, v.y);
        const float lse = m + logf(expf(v.x - m) + expf(v.y - m));
        const bool one = y[i] != 0;
        const float wi = one ? w1 : w0;
        sl = fmaf(wi, lse - (one ? v.y : v.x), sl);
        sw += wi;
    }
    sl = block_sum(sl, red);
    sw = block_sum(sw, red);
    if (threadIdx.x == 0) loss[0] = sl / sw;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float2 v = reinterpret_cast<const float2*>(x)[i];
        const float m = fmaxf(v.x, v.y);
        const float e0 = expf(v.x - m), e1 = expf(v.y - m);
        const float p1 = e1 / (e0 + e1);
        const bool one = y[i] != 0;
        const float k = (one ? w1 : w0) / sw;
        const float g1 = k * (p1 - (one ? 1.f : 0.f));
        reinterpret_cast<float2*>(dx)[i] = make_float2(-g1, g1);
    }
}

}  // namespace

extern "C" {

int seam_ce2_fwd_bwd_f32(const float* logits, const int64_t* target, const float* weight, float* loss, float* dlogits,
                         int64_t n, void* stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(ce2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, weight, loss, dlogits, (long)n);
    return (int)hipGetLastError();
}

static void wgrad_plan(int M, int C, int K, int R, int S, int& tiles_k, int& tiles_c, int& nchunks, int& cps, int& splits) {
    tiles_k = (K + 127) / 128;
    tiles_c = (C + 127) / 128;
    nchunks = (M + 31) / 32;
    const int tiles = tiles_k * tiles_c * R * S;
    splits = (1024 + tiles - 1) / tiles;
    if (splits > nchunks) splits = nchunks;
    if (splits < 1) splits = 1;
    cps = (nchunks + splits - 1) / splits;
    splits = (nchunks + cps - 1) / cps;
}

int64_t seam_conv_wgrad_workspace_floats(int M, int C, int K, int R, int S) {
    int tk, tc, nch, cps, splits;
    wgrad_plan(M, C, K, R, S, tk, tc, nch, cps, splits);
    return (int64_t)splits * R * S * K * C;
}

int seam_conv_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                        int stride, int pad, float* ws, void* stream) {
    WgradArgs a;
    a.x = x; a.dy = dy; a.ws = ws;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
    a.Ho = (H + 2 * pad - R) / stride + 1;
    a.Wo = (W + 2 * pad - S) / stride + 1;
    if (N <= 0 || a.Ho <= 0 || a.Wo <= 0 || (C % 4) || (K % 4)) return (int)hipErrorInvalidValue;
    a.M = N * a.Ho * a.Wo;
    if ((double)a.M * K * 4 >= 2147483648.0 || (double)N * H * W * C * 4 >= 2147483648.0) return (int)hipErrorInvalidValue;
    int splits;
    wgrad_plan(a.M, C, K, R, S, a.tiles_k, a.tiles_c, a.nchunks, a.chunks_per_split, splits);
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(a.tiles_k * a.tiles_c * R * S, splits), dim3(256), 0, (hipStream_t)stream, a);
    const size_t per = (size_t)R * S * K * C;
    int grid = (int)((per + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ws, dw, splits, R * S, K, C);
    return (int)hipGetLastError();
}

static int colsum_slabs(int M) {
    int slabs = (M + 127) / 128;
    return slabs > 128 ? 128 : (slabs < 1 ? 1 : slabs);
}

int64_t seam_colsum_workspace_floats(int M, int K) { return (int64_t)colsum_slabs(M) * K; }

int seam_colsum_f32(const float* x, float* out, int M, int K, float* ws, void* stream) {
    if (K <= 0) return 0;
    const int slabs = colsum_slabs(M);
    const int rows_per = (M + slabs - 1) / slabs;
    hipLaunchKernelGGL(colsum_kernel, dim3((K + 63) / 64, slabs), dim3(256), 0, (hipStream_t)stream, x, ws, M, K, rows_per);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, out, slabs, K);
    return (int)hipGetLastError();
}

int seam_avgpool_relu_bwd_f32(const float* dpool, const float* y, float* dy, int N, int HW, int C, void* stream) {
    const size_t total = (size_t)N * HW * C;
    if (total == 0) return 0;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(avgpool_relu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dpool, y, dy, HW, C, total);
    return (int)hipGetLastError();
}

int seam_bn1d_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                            float* save_invstd, float* running_mean, float* running_var, int M, int F, float momentum,
                            float eps, void* stream) {
    if (M < 2 || F <= 0) return (int)hipErrorInvalidValue;      // torch: "Expected more than 1 value per channel when training"
    hipLaunchKernelGGL(bn1d_train_fwd_kernel, dim3((F + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, gamma, beta, y,
                       save_mean, save_invstd, running_mean, running_var, M, F, momentum, eps);
    return (int)hipGetLastError();
}

int seam_bn1d_bwd_f32(const float* dy, const float* x, const float* save_mean, const float* save_invstd,
                      const float* gamma, float* dx, float* dgamma, float* dbeta, int M, int F, int frozen,
                      void* stream) {
    if (M <= 0 || F <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((F + 63) / 64), dim3(64), 0, (hipStream_t)stream, dy, x, save_mean,
                       save_invstd, gamma, dx, dgamma, dbeta, M, F, frozen);
    return (int)hipGetLastError();
}

int seam_pair_logits_bwd_f32(const float* a, const float* b, const float* w, const float* g, float* da, float* db,
                             float* dw, float* dbias, int Q, int G, int Dd, void* stream) {
    if (Dd != D) return (int)hipErrorInvalidValue;
    if (Q <= 0 || G <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pair_bwd_ab_kernel, dim3(Q + G), dim3(256), 0, (hipStream_t)stream, a, b, w, g, da, db, Q, G);
    hipLaunchKernelGGL(pair_bwd_w_kernel, dim3(D + 1), dim3(256), 0, (hipStream_t)stream, a, b, g, dw, dbias, Q, G);
    return (int)hipGetLastError();
}

int64_t seam_nlb_bwd_workspace_floats(int S, int Tmax) {
    return (int64_t)S * Tmax * (DI * 4 + D * 2 + 3) + 2 * D + 2 + 16 + D + 16;
}

static int nlb_bwd_launch(const float* seq, int64_t t_stride, int64_t s_stride, const int* len, int S, int Tmax,
                          const float* w_proj_t, const float* b_proj, const float* w_cat, const float* w_out_t,
                          const float* b_out, const float* w_att, const float* b_att, const float* dout,
                          const float* dz, int64_t dz_t_stride, int64_t dz_s_stride,
                          float* dseq, float* const* grads, int n_grads, float* ws, int use_nlb, void* stream);

// Gradients of seam_nlb_attnpool_f32 (same operand layouts).  grads: 11 output pointers in the reference's parameter
// layouts: theta.weight [128,256], theta.bias [128], phi.weight, phi.bias, g.weight, g.bias, concat_project [256],
// W.weight [256,128], W.bias [256], attention_scorer.weight [256], attention_scorer.bias [1].
int seam_nlb_attnpool_bwd_f32(const float* seq, int64_t t_stride, int64_t s_stride, const int* len, int S, int Tmax,
                              const float* w_proj_t, const float* b_proj, const float* w_cat, const float* w_out_t,
                              const float* b_out, const float* w_att, const float* b_att, const float* dout,
                              float* dseq, float* const* grads, float* ws, int use_nlb, void* stream) {
    return nlb_bwd_launch(seq, t_stride, s_stride, len, S, Tmax, w_proj_t, b_proj, w_cat, w_out_t, b_out, w_att, b_att, dout,
                          nullptr, 0, 0, dseq, grads, 11, ws, use_nlb, stream);
}

// Gradients of the non-local block ALONE (seam_nlb_attnpool_f32's z output; ref models/nlb.py:66-101 called directly):
// dz rows [s*dz_s_stride + t*dz_t_stride][256] -> dseq and grads[9] (the first nine of the list above).
int seam_nlb_block_bwd_f32(const float* seq, int64_t t_stride, int64_t s_stride, const int* len, int S, int Tmax,
                           const float* w_proj_t, const float* b_proj, const float* w_cat, const float* w_out_t,
                           const float* b_out, const float* dz, int64_t dz_t_stride, int64_t dz_s_stride,
                           float* dseq, float* const* grads, float* ws, int use_nlb, void* stream) {
    if (dz == nullptr) return (int)hipErrorInvalidValue;
    return nlb_bwd_launch(seq, t_stride, s_stride, len, S, Tmax, w_proj_t, b_proj, w_cat, w_out_t, b_out, nullptr, nullptr, nullptr,
                          dz, dz_t_stride, dz_s_stride, dseq, grads, 9, ws, use_nlb, stream);
}

static int nlb_bwd_launch(const float* seq, int64_t t_stride, int64_t s_stride, const int* len, int S, int Tmax,
                          const float* w_proj_t, const float* b_proj, const float* w_cat, const float* w_out_t,
                          const float* b_out, const float* w_att, const float* b_att, const float* dout,
                          const float* dz, int64_t dz_t_stride, int64_t dz_s_stride,
                          float* dseq, float* const* grads, int n_grads, float* ws, int use_nlb, void* stream) {
    if (S <= 0) return 0;
    if (Tmax > TB || Tmax <= 0) return (int)hipErrorInvalidValue;
    NlbParamArgs pa;
    NlbBwdArgs& a = pa.b;
    a.seq = seq; a.t_stride = t_stride; a.s_stride = s_stride; a.len = len; a.S = S; a.Tmax = Tmax;
    a.w_proj_t = w_proj_t; a.b_proj = b_proj; a.w_cat = w_cat; a.w_out_t = w_out_t; a.b_out = b_out;
    a.w_att = w_att; a.b_att = b_att; a.dout = dout; a.dseq = dseq; a.use_nlb = use_nlb;
    a.dz_ext = dz; a.dz_t_stride = dz_t_stride; a.dz_s_stride = dz_s_stride;
    const size_t rows = (size_t)S * Tmax;
    float* q = ws;
    a.G = q; q += rows * DI;
    a.Y = q; q += rows * DI;
    a.dY = q; q += rows * DI;
    a.dG = q; q += rows * DI;
    a.Z = q; q += rows * D;
    a.dZn = q; q += rows * D;
    a.vec = q; q += rows * 3;
    pa.tmp = q; q += 2 * D + 2 + 16;
    pa.d_theta_w = grads[0]; pa.d_theta_b = grads[1]; pa.d_phi_w = grads[2]; pa.d_phi_b = grads[3];
    pa.d_g_w = grads[4]; pa.d_g_b = grads[5]; pa.d_cat = grads[6]; pa.d_W_w = grads[7]; pa.d_W_b = grads[8];
    if (n_grads >= 11) { pa.d_att_w = grads[9]; pa.d_att_b = grads[10]; }
    else { pa.d_att_w = q; pa.d_att_b = q + D; }          // no scorer behind a direct block call: zeros land in scratch
    hipLaunchKernelGGL(nlb_bwd_seq_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(nlb_param_grad_kernel, dim3(DI + D + 1), dim3(256), 0, (hipStream_t)stream, pa);
    hipLaunchKernelGGL(nlb_param_assemble_kernel, dim3(2 * DI), dim3(256), 0, (hipStream_t)stream, pa);
    return (int)hipGetLastError();
}

}  // extern "C"
