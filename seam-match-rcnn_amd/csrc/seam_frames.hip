// seam_frames.hip -- the step in front of the path: what MovingFashionDataset.__getitem__ does to a
// decoded video frame on the CPU (ref datasets/MFDataset.py:79-93) as device kernels on the uint8 frame:
//   BGR -> RGB, additive Gaussian noise in [0,1] units (float64, as NumPy computes it), clip, truncate to uint8,
//   then PIL's Image.resize to half resolution.  Image.resize defaults to BICUBIC with an antialiasing support scaled
//   by the reduction factor and 8-bit fixed-point coefficients (Pillow src/libImaging/Resample.c: precompute_coeffs,
//   normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc); the kernels below restate that arithmetic
//   operation for operation (float64 coefficient maths with contraction off, 22-bit fixed point, horizontal pass
//   rounded to uint8 before the vertical pass) so the result is bit-identical to Pillow's.
// ToTensor (/255) and GeneralizedRCNNTransform are fused further down the line in seam_preprocess_u8.
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma STDC FP_CONTRACT OFF

namespace {

constexpr int KMAX = 24;               // taps per output sample: 2*ceil(2*scale)+1  (scale <= 5.5)
constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// out[y][x][c] = uint8(clip((in[y][x][2-c] / 255.0 + n * sigma) * 255.0, 0, 255))     (ref MFDataset.py:81-88)
// n = noise[y][x][c] (float64 standard normal draws, e.g. np.random.randn) or, when noise == NULL, a counter-based
// Box-Muller draw keyed by (seed, element index).  sigma == 0 and noise == NULL: plain channel flip.
__global__ void frame_noise_kernel(const uint8_t* __restrict__ in, const double* __restrict__ noise, uint8_t* __restrict__ out,
                                   size_t n_elem, double sigma, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elem; i += (size_t)gridDim.x * blockDim.x) {
        const size_t px = i / 3;
        const int c = (int)(i - px * 3);
        const uint8_t v = in[px * 3 + (2 - c)];
        if (sigma == 0.0 && noise == nullptr) { out[i] = v; continue; }
        double n;
        if (noise) {
            n = noise[i];
        } else {
            const uint64_t r0 = splitmix64(seed ^ (i * 2 + 1)), r1 = splitmix64(seed ^ (i * 2 + 2) ^ 0xD1B54A32D192ED03ull);
            const double u0 = ((double)(r0 >> 11) + 1.0) * (1.0 / 9007199254740993.0);    // (0,1)
            const double u1 = (double)(r1 >> 11) * (1.0 / 9007199254740992.0);
            n = sqrt(-2.0 * log(u0)) * cos(6.283185307179586 * u1);
        }
        double x = (double)v / 255.0;
        x = x + n * sigma;
        x = x * 255.0;
        x = x < 0.0 ? 0.0 : (x > 255.0 ? 255.0 : x);
        out[i] = (uint8_t)x;
    }
}

__device__ __forceinline__ double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for one axis: bounds[xx] = (xmin, count), kk[xx][KMAX] fixed point.
__global__ void resize_coeff_kernel(int inSize, int outSize, int* __restrict__ bounds, int* __restrict__ kk) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= outSize) return;
    const float in0 = 0.f, in1 = (float)inSize;
    double scale = (double)(in1 - in0) / outSize;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const double center = in0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    double k[KMAX];
    double ww = 0.0;
    for (int x = 0; x < KMAX; ++x) {
        double w = 0.0;
        if (x < xmax) {
            w = bicubic_filter((x + xmin - center + 0.5) * ss);
            ww += w;
        }
        k[x] = w;
    }
    for (int x = 0; x < KMAX; ++x) {
        double v = k[x];
        if (x < xmax && ww != 0.0) v /= ww;
        kk[(size_t)xx * KMAX + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one resampling pass over an interleaved uint8 image [H][W][3]; axis 0: along x (out [H][OW][3]), axis 1: along y
__global__ void resize_pass_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int OH, int OW,
                                   const int* __restrict__ bounds, const int* __restrict__ kk, int axis) {
    const size_t total = (size_t)OH * OW * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const int x = (int)((i / 3) % OW);
        const int y = (int)(i / ((size_t)3 * OW));
        const int o = axis == 0 ? x : y;
        const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
        const int* k = kk + (size_t)o * KMAX;
        int ss = 1 << (PRECISION_BITS - 1);
        if (axis == 0) {
            const uint8_t* row = in + ((size_t)y * W + lo) * 3 + c;
            for (int t = 0; t < cnt; ++t) ss += (int)row[(size_t)t * 3] * k[t];
        } else {
            const uint8_t* col = in + ((size_t)lo * W + x) * 3 + c;
            for (int t = 0; t < cnt; ++t) ss += (int)col[(size_t)t * W * 3] * k[t];
        }
        out[i] = clip8(ss);
    }
}

inline int taps_needed(int inSize, int outSize) {
    double fs = (double)inSize / outSize;
    if (fs < 1.0) fs = 1.0;
    const double support = 2.0 * fs;
    int c = (int)support;
    if ((double)c < support) ++c;
    return c * 2 + 1;
}

}  // namespace

extern "C" {

int seam_frame_noise_u8(const uint8_t* bgr, const double* noise, uint8_t* rgb, int H, int W, double sigma, uint64_t seed,
                        void* stream) {
    const size_t n = (size_t)H * W * 3;
    if (n == 0) return 0;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(frame_noise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, bgr, noise, rgb, n, sigma, seed);
    return (int)hipGetLastError();
}

// bytes of scratch for seam_resize_bicubic_u8: coefficient tables of both axes + the horizontally resampled image
int64_t seam_resize_workspace_bytes(int H, int W, int OH, int OW) {
    return (int64_t)(OW + OH) * (KMAX + 2) * 4 + (int64_t)H * OW * 3 + 64;
}

int seam_resize_bicubic_u8(const uint8_t* in, uint8_t* out, int H, int W, int OH, int OW, void* ws, void* stream) {
    if (H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return (int)hipErrorInvalidValue;
    if (taps_needed(W, OW) > KMAX || taps_needed(H, OH) > KMAX) return (int)hipErrorInvalidValue;
    int* bx = (int*)ws;
    int* kx = bx + 2 * OW;
    int* by = kx + (size_t)OW * KMAX;
    int* ky = by + 2 * OH;
    uint8_t* tmp = (uint8_t*)(ky + (size_t)OH * KMAX);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(resize_coeff_kernel, dim3((OW + 255) / 256), dim3(256), 0, s, W, OW, bx, kx);
    hipLaunchKernelGGL(resize_coeff_kernel, dim3((OH + 255) / 256), dim3(256), 0, s, H, OH, by, ky);
    auto grid = [](size_t n) { size_t g = (n + 255) / 256; return dim3((unsigned)(g > 16384 ? 16384 : g)); };
    // Pillow resamples horizontally first (ImagingResample), rounding to uint8 between the passes
    hipLaunchKernelGGL(resize_pass_kernel, grid((size_t)H * OW * 3), dim3(256), 0, s, in, tmp, H, W, H, OW, bx, kx, 0);
    hipLaunchKernelGGL(resize_pass_kernel, grid((size_t)OH * OW * 3), dim3(256), 0, s, tmp, out, H, OW, OH, OW, by, ky, 1);
    return (int)hipGetLastError();
}

}  // extern "C"
