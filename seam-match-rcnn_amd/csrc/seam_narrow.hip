// seam_narrow.hip -- 1x1 convolutions / Linear layers with at most 16 outputs (RPN objectness + box deltas: 15 per pixel
// [TV RPNHead.cls_logits / bbox_pred]; mask logits: 14 per sub-pixel [TV MaskRCNNPredictor.mask_fcn_logits]).
//
// These layers are HBM-bound (256 input floats per 15 outputs: AI = 7.5 FLOP/B), but on the 128x64 tile of conv_igemm
// they are bound by the MFMAs of 49 padded output columns (20-25 TFLOP/s algorithmic).  Here a wave owns 16 rows at a
// time and multiplies them with v_mfma_f32_16x16x4_f32 (one 16-column tile: no padding work to speak of); the whole
// [16 x C] weight matrix lives in registers (C <= 256: 64 VGPRs), the rows stream through two register sets with
// 16-byte loads (a lane takes channels 16 j + 4 (l >> 4) + e of row l & 15; the same k permutation is baked into the
// packed weights), so the kernel is a pure row stream: x is read once, nothing else moves.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kOob = 0x80000000u;
constexpr int JMAX = 16;      // 16-channel blocks per row: C <= 256

__global__ __launch_bounds__(256, 2) void linear_narrow_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int M, int C, int K, int relu) {
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, kq = lane >> 4;
    const int cb = C >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const long ntiles = ((long)M + 15) >> 4;

    f32x4 b[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        b[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (j < cb) b[j] = *reinterpret_cast<const f32x4*>(wpk + ((size_t)j * 64 + lane) * 4);
    }
    const float bv = (bias && col < K) ? bias[col] : 0.f;
    const int row_bytes = C * 4;
    const int lane_off = kq * 16;                         // bytes inside a 64-byte channel block

    f32x4 a0[JMAX], a1[JMAX];
    auto load_tile = [&](f32x4 (&a)[JMAX], long tile) {
        // descriptor rebased at the tile's first row: offsets stay small for any M
        const long row0 = tile << 4;
        const long rows_left = (long)M - row0;
        const unsigned nrec = rows_left <= 0 ? 0u : (unsigned)((rows_left < 16 ? rows_left : 16) * row_bytes);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)x + (size_t)(rows_left > 0 ? row0 : 0) * row_bytes),
                                                                            0, (int)nrec, 0x00020000);
        const unsigned base = (unsigned)(col * row_bytes + lane_off);
#pragma unroll
        for (int j = 0; j < JMAX; ++j)
            a[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, j < cb ? base + j * 64 : kOob, 0, 0));
    };
    auto compute_store = [&](const f32x4 (&a)[JMAX], long tile) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < JMAX; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][e], b[j][e], acc, 0, 0, 0);
        const long row0 = (tile << 4) + 4 * kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[i] + bv;
            if (relu) v = fmaxf(v, 0.f);
            const long row = row0 + i;
            if (col < K && row < M) y[row * K + col] = v;
        }
    };

    long t = wave;
    if (t < ntiles) load_tile(a0, t);
    while (t < ntiles) {
        const long t1 = t + nwaves;
        if (t1 < ntiles) load_tile(a1, t1);
        compute_store(a0, t);
        t = t1;
        if (t >= ntiles) break;
        const long t2 = t + nwaves;
        if (t2 < ntiles) load_tile(a0, t2);
        compute_store(a1, t);
        t = t2;
    }
}

// [K, C] fp32 row-major -> [C/16][64 lanes][4]: lane (col = l & 15, kq = l >> 4), element e = w[col][16 j + 4 kq + e] (0 for col >= K)
__global__ void narrow_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int K, int C) {
    const int total = (C >> 4) * 64 * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i & 3, lane = (i >> 2) & 63, j = i >> 8;
        const int col = lane & 15, kq = lane >> 4;
        out[i] = col < K ? w[(size_t)col * C + 16 * j + 4 * kq + e] : 0.f;
    }
}

}  // namespace

extern "C" {

int seam_linear_narrow_supported(int C, int K) { return (C % 16 == 0 && C >= 16 && C <= 16 * JMAX && K >= 1 && K <= 16) ? 1 : 0; }

int seam_pack_linear_narrow_f32(const float* w, float* w_packed, int K, int C, void* stream) {
    if (!seam_linear_narrow_supported(C, K)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(narrow_pack_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w, w_packed, K, C);
    return (int)hipGetLastError();
}

int seam_linear_narrow_f32(const float* x, const float* w_packed, const float* bias, float* y, long long M, int C, int K, int relu,
                           void* stream) {
    if (!seam_linear_narrow_supported(C, K) || M <= 0 || M > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const long long tiles = (M + 15) / 16;
    long long blocks = (tiles + 3) / 4;
    if (blocks > 512) blocks = 512;                   // two blocks per CU, each wave walks tiles
    hipLaunchKernelGGL(linear_narrow_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w_packed, bias, y, (int)M, C, K,
                       relu);
    return (int)hipGetLastError();
}

}  // extern "C"
