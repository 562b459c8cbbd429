// seam_elementwise.hip -- HBM-bound glue kernels of the path (NHWC, 16 B per lane), fp32 and fp16:
//   preprocess   GeneralizedRCNNTransform: normalise + bilinear resize + pad + CHW->NHWC (4 / 8 stored ch.)
//   maxpool2d    ResNet stem pool / FPN LastLevelMaxPool
//   upsample_add FPN top-down merge (nearest)
//   transposes   NCHW <-> NHWC bridges at the module boundary (LDS-tiled, both sides coalesced)
//   avgpool      AvgPool2d((6,6)) of the match trunk
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

template <typename T> struct Vec16;                       // 16 bytes of T
template <> struct Vec16<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };
template <> struct Vec16<_Float16> { typedef _Float16 type __attribute__((ext_vector_type(8))); static constexpr int N = 8; };

__device__ __forceinline__ void bilinear_axis(int dst, int in, int out, int& i0, int& i1, float& l1) {
    // ATen upsample_bilinear2d, align_corners=False, scale = in/out (recompute_scale_factor=True)
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    int i = (int)src;
    if (i > in - 1) i = in - 1;
    i0 = i;
    i1 = (i < in - 1) ? i + 1 : i;
    float l = src - (float)i;
    l1 = fminf(fmaxf(l, 0.f), 1.f);
}

template <typename T>
__global__ void preprocess_kernel(const float* __restrict__ img, T* __restrict__ out, int in_h, int in_w,
                                  int out_h, int out_w, int Hp, int Wp, size_t img_stride) {
    typedef typename Vec16<T>::type V;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= Wp) return;
    img += (size_t)blockIdx.z * img_stride;                                // image blockIdx.z of a same-shape batch
    out += (size_t)blockIdx.z * Hp * Wp * Vec16<T>::N;
    float v[3] = {0.f, 0.f, 0.f};
    if (y < out_h && x < out_w) {
        const float mean[3] = {0.485f, 0.456f, 0.406f};
        const float stdv[3] = {0.229f, 0.224f, 0.225f};
        if (out_h == in_h && out_w == in_w) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (img[((size_t)c * in_h + y) * in_w + x] - mean[c]) / stdv[c];
        } else {
            int y0, y1, x0, x1;
            float ly, lx;
            bilinear_axis(y, in_h, out_h, y0, y1, ly);
            bilinear_axis(x, in_w, out_w, x0, x1, lx);
            const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* p = img + (size_t)c * in_h * in_w;
                const float v00 = (p[(size_t)y0 * in_w + x0] - mean[c]) / stdv[c];
                const float v01 = (p[(size_t)y0 * in_w + x1] - mean[c]) / stdv[c];
                const float v10 = (p[(size_t)y1 * in_w + x0] - mean[c]) / stdv[c];
                const float v11 = (p[(size_t)y1 * in_w + x1] - mean[c]) / stdv[c];
                v[c] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
            }
        }
    }
    V o;
#pragma unroll
    for (int c = 0; c < Vec16<T>::N; ++c) o[c] = (T)(c < 3 ? v[c] : 0.f);
    *reinterpret_cast<V*>(out + ((size_t)y * Wp + x) * Vec16<T>::N) = o;
}

// The same transform written space-to-depth: out [N, Hp/2, Wp/2, 12], channel (dy*2 + dx)*3 + c = pixel (2Y+dy, 2X+dx), colour c.
// The 7x7 / stride-2 stem then is a 4x4 / stride-1 convolution over 12 channels (192 reduction steps = 6 chunks exactly, no
// padded 4th colour channel: the NHWC4 form spends 224 steps on 147 products) whose gathers are 48-byte pixels.
template <typename T>
__global__ void preprocess_s2d_kernel(const float* __restrict__ img, T* __restrict__ out, int in_h, int in_w,
                                      int out_h, int out_w, int Hp, int Wp, size_t img_stride, int pad_lo, int pad_hi) {
    // pad_lo / pad_hi (round 6): zero cells written before / after the frame in both directions -- the grid covers the padded frame
    // [H2 + pad_lo + pad_hi, W2 + pad_lo + pad_hi]; cells outside the frame proper fall through the (y, x) range test below as zeros
    const int Xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int Yo = blockIdx.y;
    const int W2 = Wp >> 1, H2 = Hp >> 1;
    const int W2p = W2 + pad_lo + pad_hi, H2p = H2 + pad_lo + pad_hi;
    if (Xo >= W2p) return;
    const int X = Xo - pad_lo, Y = Yo - pad_lo;
    const bool inside = X >= 0 && X < W2 && Y >= 0 && Y < H2;
    img += (size_t)blockIdx.z * img_stride;
    float o[12];
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    // no resize, even row length, 8-byte aligned planes (the 1080p / 800^2 frames of the configs): the two pixels of a cell row are ONE
    // 8-byte load per colour plane -- half the load instructions of the general form below, the same arithmetic on the same values
    const bool pair = out_h == in_h && out_w == in_w && !(in_w & 1) && !(img_stride & 1) && !((size_t)img & 7);
    if (pair) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = 2 * Y + r;
            f32x2 v2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            if (inside && y < out_h && 2 * X < out_w) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v2[c] = *reinterpret_cast<const f32x2*>(img + ((size_t)c * in_h + y) * in_w + 2 * X);
#pragma unroll
                for (int c = 0; c < 3; ++c) { v2[c][0] = (v2[c][0] - mean[c]) / stdv[c]; v2[c][1] = (v2[c][1] - mean[c]) / stdv[c]; }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { o[(2 * r) * 3 + c] = v2[c][0]; o[(2 * r + 1) * 3 + c] = v2[c][1]; }
        }
    } else
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int y = 2 * Y + (d >> 1), x = 2 * X + (d & 1);
        float v[3] = {0.f, 0.f, 0.f};
        if (inside && y < out_h && x < out_w) {
            if (out_h == in_h && out_w == in_w) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = (img[((size_t)c * in_h + y) * in_w + x] - mean[c]) / stdv[c];
            } else {
                int y0, y1, x0, x1;
                float ly, lx;
                bilinear_axis(y, in_h, out_h, y0, y1, ly);
                bilinear_axis(x, in_w, out_w, x0, x1, lx);
                const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* p = img + (size_t)c * in_h * in_w;
                    const float v00 = (p[(size_t)y0 * in_w + x0] - mean[c]) / stdv[c];
                    const float v01 = (p[(size_t)y0 * in_w + x1] - mean[c]) / stdv[c];
                    const float v10 = (p[(size_t)y1 * in_w + x0] - mean[c]) / stdv[c];
                    const float v11 = (p[(size_t)y1 * in_w + x1] - mean[c]) / stdv[c];
                    v[c] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) o[d * 3 + c] = v[c];
    }
    if constexpr (sizeof(T) == 2) {             // fp16: 16 channels per cell (12 + 4 zeros: the fp16 GEMM reads 8-channel vectors), 32 bytes
        typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
        T* dst = out + (((size_t)blockIdx.z * H2p + Yo) * W2p + Xo) * 16;
        h16x8 a, b;
#pragma unroll
        for (int c = 0; c < 8; ++c) { a[c] = (_Float16)o[c]; b[c] = (_Float16)(c < 4 ? o[8 + c] : 0.f); }
        *reinterpret_cast<h16x8*>(dst) = a;
        *reinterpret_cast<h16x8*>(dst + 8) = b;
    } else {
        T* dst = out + (((size_t)blockIdx.z * H2p + Yo) * W2p + Xo) * 12;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            f32x4 v4 = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
            *reinterpret_cast<f32x4*>(dst + 4 * q) = v4;
        }
    }
}

// uint8 HWC RGB frame -> ToTensor (/255, ref stuffs/transform.py:46-49) + the transform above, one pass:
// the clip crosses PCIe as 1 byte per sample instead of 4 (row f4, device side).
template <typename T>
__global__ void preprocess_u8_kernel(const uint8_t* __restrict__ img, T* __restrict__ out, int in_h, int in_w,
                                     int out_h, int out_w, int Hp, int Wp) {
    typedef typename Vec16<T>::type V;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= Wp) return;
    float v[3] = {0.f, 0.f, 0.f};
    if (y < out_h && x < out_w) {
        const float mean[3] = {0.485f, 0.456f, 0.406f};
        const float stdv[3] = {0.229f, 0.224f, 0.225f};
        auto px = [&](int yy, int xx, int c) -> float {
            return ((float)img[((size_t)yy * in_w + xx) * 3 + c] / 255.f - mean[c]) / stdv[c];
        };
        if (out_h == in_h && out_w == in_w) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = px(y, x, c);
        } else {
            int y0, y1, x0, x1;
            float ly, lx;
            bilinear_axis(y, in_h, out_h, y0, y1, ly);
            bilinear_axis(x, in_w, out_w, x0, x1, lx);
            const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[c] = hy * (hx * px(y0, x0, c) + lx * px(y0, x1, c)) + ly * (hx * px(y1, x0, c) + lx * px(y1, x1, c));
        }
    }
    V o;
#pragma unroll
    for (int c = 0; c < Vec16<T>::N; ++c) o[c] = (T)(c < 3 ? v[c] : 0.f);
    *reinterpret_cast<V*>(out + ((size_t)y * Wp + x) * Vec16<T>::N) = o;
}

template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                               int Ho, int Wo, int k, int stride, int pad) {
    typedef typename Vec16<T>::type V;
    constexpr int E = Vec16<T>::N;
    const int cv = C / E;
    const size_t total = (size_t)N * Ho * Wo * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        size_t r = i / cv;
        const int wo = (int)(r % Wo);
        r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float m[E];
#pragma unroll
        for (int e = 0; e < E; ++e) m[e] = -INFINITY;
        for (int a = 0; a < k; ++a) {
            const int hi = ho * stride - pad + a;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int b = 0; b < k; ++b) {
                const int wi = wo * stride - pad + b;
                if ((unsigned)wi >= (unsigned)W) continue;
                const V v = *reinterpret_cast<const V*>(x + (((size_t)n * H + hi) * W + wi) * C + c * E);
#pragma unroll
                for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], (float)v[e]);
            }
        }
        V o;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = (T)m[e];
        *reinterpret_cast<V*>(y + i * E) = o;
    }
}

// 3 x 3 / stride 2 / pad 1 (ResNet's pool after the stem -- the one large max-pool of the path): one thread = one output pixel x 16
// bytes of channels; window coordinates CLAMPED into the map instead of tested (a clamped tap repeats a tap of the window: the
// maximum is the same), 32-bit index arithmetic, nine independent 16-byte loads.  Same values as maxpool_kernel.
template <typename T>
__global__ void maxpool3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int Ho, int Wo, unsigned m_cv, unsigned m_Wo) {
    typedef typename Vec16<T>::type V;
    constexpr int E = Vec16<T>::N;
    const int cv = C / E;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;           // (ho, wo, c) of image blockIdx.y
    if (i >= (unsigned)(Ho * Wo * cv)) return;
    const unsigned pix = cv == 1 ? i : __umulhi(i, m_cv);
    const int c = (int)(i - pix * cv);
    const int ho = (int)(Wo == 1 ? pix : __umulhi(pix, m_Wo));
    const int wo = (int)pix - ho * Wo;
    const T* xi = x + (size_t)blockIdx.y * H * W * C + c * E;
    int hs[3], ws[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        hs[a] = min(max(2 * ho - 1 + a, 0), H - 1);
        ws[a] = min(max(2 * wo - 1 + a, 0), W - 1);
    }
    V v[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) v[a * 3 + b] = *reinterpret_cast<const V*>(xi + (size_t)(hs[a] * W + ws[b]) * C);
    V o;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float m = (float)v[0][e];
#pragma unroll
        for (int t = 1; t < 9; ++t) m = fmaxf(m, (float)v[t][e]);
        o[e] = (T)m;
    }
    *reinterpret_cast<V*>(y + ((size_t)blockIdx.y * Ho * Wo * cv + i) * E) = o;
}

template <typename T>
__global__ void upsample_add_kernel(T* __restrict__ lat, const T* __restrict__ top, int N, int H, int W,
                                    int Ht, int Wt, int C) {
    typedef typename Vec16<T>::type V;
    constexpr int E = Vec16<T>::N;
    const int cv = C / E;
    const size_t total = (size_t)N * H * W * cv;
    const float sh = (float)Ht / (float)H, sw = (float)Wt / (float)W;   // ATen nearest: floor(dst*scale)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        size_t r = i / cv;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        int ht = (int)floorf((float)h * sh);
        int wt = (int)floorf((float)w * sw);
        if (ht > Ht - 1) ht = Ht - 1;
        if (wt > Wt - 1) wt = Wt - 1;
        V a = *reinterpret_cast<V*>(lat + i * E);
        const V b = *reinterpret_cast<const V*>(top + (((size_t)n * Ht + ht) * Wt + wt) * C + c * E);
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] = (T)((float)a[e] + (float)b[e]);
        *reinterpret_cast<V*>(lat + i * E) = a;
    }
}

// y[b][j][i] = x[b][i][j]; x is [B][rows][cols]
template <typename TI, typename TO>
__global__ void transpose_kernel(const TI* __restrict__ x, TO* __restrict__ y, int rows, int cols) {
    __shared__ float tile[32][33];
    const size_t boff = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int r = r0 + ty + k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + k][tx] = (float)x[boff + (size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int c = c0 + ty + k, r = r0 + tx;
        if (r < rows && c < cols) y[boff + (size_t)c * rows + r] = (TO)tile[tx][ty + k];
    }
}

template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int K, int L, int C) {
    const size_t total = (size_t)K * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t k = i / C;
        const T* p = x + k * L * C + c;
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += (float)p[(size_t)l * C];
        y[i] = (T)(s / (float)L);
    }
}

// the same sums in the same order (l = 0 .. L - 1 in fp32, one division), 16 bytes of channels per thread: the scalar form above
// moves 2-4 bytes per lane and load (the 6 x 6 x 1024 trunk outputs of a config-5 step: 1.9 TB/s)
template <typename T>
__global__ void avgpool_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int K, int L, int C) {
    typedef typename Vec16<T>::type V;
    constexpr int E = Vec16<T>::N;
    const int cv = C / E;
    const size_t total = (size_t)K * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        const size_t k = i / cv;
        const T* p = x + k * L * C + c * E;
        float s[E];
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] = 0.f;
        int l = 0;
        for (; l + 4 <= L; l += 4) {
            V v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const V*>(p + (size_t)(l + u) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < E; ++e) s[e] += (float)v[u][e];
        }
        for (; l < L; ++l) {
            const V v = *reinterpret_cast<const V*>(p + (size_t)l * C);
#pragma unroll
            for (int e = 0; e < E; ++e) s[e] += (float)v[e];
        }
        V o;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = (T)(s[e] / (float)L);
        *reinterpret_cast<V*>(y + i * E) = o;
    }
}

inline int grid_for(size_t total, int block = 256, int cap = 256 * 16) {
    size_t g = (total + block - 1) / block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

template <typename T>
int preprocess(const float* img, void* out, int in_h, int in_w, int out_h, int out_w, int Hp, int Wp, void* stream,
               int n = 1, size_t img_stride = 0) {
    if (n < 1 || n > 65535) return (int)hipErrorInvalidValue;
    dim3 grid((Wp + 255) / 256, Hp, n);
    hipLaunchKernelGGL(preprocess_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, img, (T*)out, in_h, in_w, out_h,
                       out_w, Hp, Wp, img_stride);
    return (int)hipGetLastError();
}

template <typename T>
int maxpool(const void* x, void* y, int N, int H, int W, int C, int k, int stride, int pad, void* stream) {
    if (C % Vec16<T>::N) return (int)hipErrorInvalidValue;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / Vec16<T>::N);
    const size_t per_img = (size_t)Ho * Wo * (C / Vec16<T>::N);
    if (k == 3 && stride == 2 && pad == 1 && H >= 2 && W >= 2 && N <= 65535 && per_img < (1u << 31) && (size_t)H * W * C < (1u << 31)) {
        const unsigned cv = (unsigned)(C / Vec16<T>::N);
        auto magic = [](unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); };
        hipLaunchKernelGGL(maxpool3s2_kernel<T>, dim3((unsigned)((per_img + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (T*)y, H, W, C, Ho, Wo, magic(cv), magic((unsigned)Wo));
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(maxpool_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, N,
                       H, W, C, Ho, Wo, k, stride, pad);
    return (int)hipGetLastError();
}

template <typename T>
int upsample_add(void* lat, const void* top, int N, int H, int W, int Ht, int Wt, int C, void* stream) {
    if (C % Vec16<T>::N) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)N * H * W * (C / Vec16<T>::N);
    hipLaunchKernelGGL(upsample_add_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (T*)lat,
                       (const T*)top, N, H, W, Ht, Wt, C);
    return (int)hipGetLastError();
}

template <typename TI, typename TO>
int transpose(const void* x, void* y, int B, int rows, int cols, void* stream) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, B);
    hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, dim3(256), 0, (hipStream_t)stream, (const TI*)x, (TO*)y, rows, cols);
    return (int)hipGetLastError();
}

template <typename T>
int avgpool(const void* x, void* y, int K, int L, int C, void* stream) {
    if (C % Vec16<T>::N == 0 && !((size_t)x & 15) && !((size_t)y & 15)) {
        hipLaunchKernelGGL(avgpool_vec_kernel<T>, dim3(grid_for((size_t)K * (C / Vec16<T>::N), 256, 256 * 64)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (T*)y, K, L, C);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(avgpool_kernel<T>, dim3(grid_for((size_t)K * C)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                       (T*)y, K, L, C);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int seam_preprocess_f32(const float* img, float* out, int in_h, int in_w, int out_h, int out_w, int Hp, int Wp, void* stream) {
    return preprocess<float>(img, out, in_h, in_w, out_h, out_w, Hp, Wp, stream);
}
int seam_preprocess_f16(const float* img, void* out, int in_h, int in_w, int out_h, int out_w, int Hp, int Wp, void* stream) {
    return preprocess<_Float16>(img, out, in_h, in_w, out_h, out_w, Hp, Wp, stream);
}

int seam_preprocess_batch_f32(const float* imgs, size_t img_stride, float* out, int n, int in_h, int in_w, int out_h, int out_w,
                              int Hp, int Wp, void* stream) {
    return preprocess<float>(imgs, out, in_h, in_w, out_h, out_w, Hp, Wp, stream, n, img_stride);
}
int seam_preprocess_batch_f16(const float* imgs, size_t img_stride, void* out, int n, int in_h, int in_w, int out_h, int out_w,
                              int Hp, int Wp, void* stream) {
    return preprocess<_Float16>(imgs, out, in_h, in_w, out_h, out_w, Hp, Wp, stream, n, img_stride);
}

int seam_preprocess_s2d_batch_f32(const float* imgs, size_t img_stride, float* out, int n, int in_h, int in_w, int out_h, int out_w,
                                  int Hp, int Wp, void* stream) {
    if (n < 1 || n > 65535 || (Hp & 1) || (Wp & 1)) return (int)hipErrorInvalidValue;
    dim3 grid((Wp / 2 + 127) / 128, Hp / 2, n);
    hipLaunchKernelGGL(preprocess_s2d_kernel<float>, grid, dim3(128), 0, (hipStream_t)stream, imgs, out, in_h, in_w, out_h, out_w,
                       Hp, Wp, img_stride, 0, 0);
    return (int)hipGetLastError();
}

int seam_preprocess_s2d_pad_batch_f16(const float* imgs, size_t img_stride, void* out, int n, int in_h, int in_w, int out_h, int out_w,
                                      int Hp, int Wp, int pad_lo, int pad_hi, void* stream) {
    if (n < 1 || n > 65535 || (Hp & 1) || (Wp & 1) || pad_lo < 0 || pad_hi < 0 || pad_lo > 8 || pad_hi > 8) return (int)hipErrorInvalidValue;
    dim3 grid((Wp / 2 + pad_lo + pad_hi + 127) / 128, Hp / 2 + pad_lo + pad_hi, n);
    hipLaunchKernelGGL(preprocess_s2d_kernel<_Float16>, grid, dim3(128), 0, (hipStream_t)stream, imgs, (_Float16*)out, in_h, in_w,
                       out_h, out_w, Hp, Wp, img_stride, pad_lo, pad_hi);
    return (int)hipGetLastError();
}

int seam_preprocess_s2d_batch_f16(const float* imgs, size_t img_stride, void* out, int n, int in_h, int in_w, int out_h, int out_w,
                                  int Hp, int Wp, void* stream) {
    return seam_preprocess_s2d_pad_batch_f16(imgs, img_stride, out, n, in_h, in_w, out_h, out_w, Hp, Wp, 0, 0, stream);
}

int seam_preprocess_u8(const uint8_t* img, void* out, int in_h, int in_w, int out_h, int out_w, int Hp, int Wp, int out_f16,
                       void* stream) {
    dim3 grid((Wp + 255) / 256, Hp);
    if (out_f16)
        hipLaunchKernelGGL(preprocess_u8_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, img, (_Float16*)out, in_h,
                           in_w, out_h, out_w, Hp, Wp);
    else
        hipLaunchKernelGGL(preprocess_u8_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, img, (float*)out, in_h, in_w,
                           out_h, out_w, Hp, Wp);
    return (int)hipGetLastError();
}

int seam_maxpool2d_f32(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int pad, void* stream) {
    return maxpool<float>(x, y, N, H, W, C, k, stride, pad, stream);
}
int seam_maxpool2d_f16(const void* x, void* y, int N, int H, int W, int C, int k, int stride, int pad, void* stream) {
    return maxpool<_Float16>(x, y, N, H, W, C, k, stride, pad, stream);
}

int seam_upsample_add_f32(float* lat, const float* top, int N, int H, int W, int Ht, int Wt, int C, void* stream) {
    return upsample_add<float>(lat, top, N, H, W, Ht, Wt, C, stream);
}
int seam_upsample_add_f16(void* lat, const void* top, int N, int H, int W, int Ht, int Wt, int C, void* stream) {
    return upsample_add<_Float16>(lat, top, N, H, W, Ht, Wt, C, stream);
}

// x [B,C,L] -> y [B,L,C]
int seam_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int L, void* stream) {
    return transpose<float, float>(x, y, B, C, L, stream);
}
int seam_nchw_f32_to_nhwc_f16(const float* x, void* y, int B, int C, int L, void* stream) {
    return transpose<float, _Float16>(x, y, B, C, L, stream);
}
// x [B,L,C] -> y [B,C,L]
int seam_nhwc_to_nchw_f32(const float* x, float* y, int B, int L, int C, void* stream) {
    return transpose<float, float>(x, y, B, L, C, stream);
}
int seam_nhwc_f16_to_nchw_f32(const void* x, float* y, int B, int L, int C, void* stream) {
    return transpose<_Float16, float>(x, y, B, L, C, stream);
}

int seam_avgpool_f32(const float* x, float* y, int K, int L, int C, void* stream) { return avgpool<float>(x, y, K, L, C, stream); }
int seam_avgpool_f16(const void* x, void* y, int K, int L, int C, void* stream) { return avgpool<_Float16>(x, y, K, L, C, stream); }

}  // extern "C"
