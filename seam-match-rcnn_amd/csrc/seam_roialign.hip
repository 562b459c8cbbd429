// seam_roialign.hip -- MultiScaleRoIAlign + roi_align(aligned=False) on NHWC pyramids (gfx950).
//
// Three kernels, bit-identical results (the same expression per sample, in the same order); A/B in profiles/r03_roialign_ab.txt:
//
// roi_align_quad_kernel (default) LDS-staged ROI tiles (what BASELINE.json's north_star names): one block per (ROI quadrant of 7 x 7
//                       bins, 64-channel quarter).  The quadrant's whole footprint (<= 16 x 16 feature pixels x 256 B = 64 KB) is
//                       read once, as coalesced 256-byte pixel segments, every thread's loads in flight together; while they fly the
//                       bilinear taps of the quadrant's 196 samples (4 weights + 4 LDS offsets) are computed ONCE per block -- the
//                       coordinate arithmetic, ~40 VALU instructions per sample, is what bounds this operator; ONE barrier; then
//                       49 bins x 16 channel groups come from LDS (conflict-free ds_read_b128: the 16 lanes of a read group hold the
//                       16 channel groups).  sampling_ratio 2, P <= 16, C % 64 == 0; a footprint larger than the window takes the
//                       gather path inside the same kernel.
// roi_align_kernel      one wave64 per output bin: lane l owns channels [4l,4l+4) (+256 per extra pass), so each of the
//                       16 bilinear taps of a bin is ONE coalesced 1 KiB wave load (channels are contiguous in NHWC);
//                       neighbouring bins/ROIs re-hit the taps in L1/L2; every lane repeats the sample arithmetic.  The general form
//                       (any sampling_ratio / P / C % 4 == 0).
// roi_align_lds_kernel  row-staged tiles: per row of bins the <= 4 feature rows its two sample rows touch go through LDS with a
//                       register-staged prefetch one bin row ahead.  Kept for the A/B: latency-bound (14 dependent stages), slower.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "seam_opts.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

struct RoiArgs {
    const void* feat[4];
    int h[4], w[4];
    float scale[4];
    int C, k_min;
    const float* rois;
    const int* levels;
    void* out;
    int K, P, sr;
};

__device__ __forceinline__ int map_level(float x1, float y1, float x2, float y2, int k_min) {
    // LevelMapper: floor(4 + log2(sqrt(area)/224) + 1e-6), clamped to [k_min, k_min+3]
    const float s = sqrtf((x2 - x1) * (y2 - y1));
    float l = floorf(4.f + log2f(s / 224.f) + 1e-6f);
    l = fminf(fmaxf(l, (float)k_min), (float)(k_min + 3));
    return (int)l - k_min;
}

// one bilinear sample, fixed operation order (both kernels): acc + (((w1 v1) + w2 v2) + w3 v3) + w4 v4, each "+ w v" one fma
__device__ __forceinline__ f32x4 sample_acc(f32x4 acc, float w1, f32x4 v1, float w2, f32x4 v2, float w3, f32x4 v3, float w4, f32x4 v4) {
    f32x4 s = w1 * v1;
    s = __builtin_elementwise_fma((f32x4){w2, w2, w2, w2}, v2, s);
    s = __builtin_elementwise_fma((f32x4){w3, w3, w3, w3}, v3, s);
    s = __builtin_elementwise_fma((f32x4){w4, w4, w4, w4}, v4, s);
    return acc + s;
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiArgs p) {
    typedef T tv4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int bin = blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per bin
    const int PP = p.P * p.P;
    if (bin >= p.K * PP) return;
    const int k = bin / PP;
    const int pb = bin - k * PP;
    const int ph = pb / p.P, pw = pb - ph * p.P;

    const float* r = p.rois + (size_t)k * 5;
    const int bidx = (int)r[0];
    const float bx1 = r[1], by1 = r[2], bx2 = r[3], by2 = r[4];
    const int lvl = p.levels ? p.levels[k] : map_level(bx1, by1, bx2, by2, p.k_min);
    // (select, not index: a dynamically indexed kernarg array would be spilled to scratch)
    const int H = lvl == 0 ? p.h[0] : lvl == 1 ? p.h[1] : lvl == 2 ? p.h[2] : p.h[3];
    const int W = lvl == 0 ? p.w[0] : lvl == 1 ? p.w[1] : lvl == 2 ? p.w[2] : p.w[3];
    const float sc = lvl == 0 ? p.scale[0] : lvl == 1 ? p.scale[1] : lvl == 2 ? p.scale[2] : p.scale[3];
    const T* fb = (const T*)(lvl == 0 ? p.feat[0] : lvl == 1 ? p.feat[1] : lvl == 2 ? p.feat[2] : p.feat[3]);
    const T* f = fb + (size_t)bidx * H * W * p.C;

    const float x1 = bx1 * sc, y1 = by1 * sc, x2 = bx2 * sc, y2 = by2 * sc;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bw = rw / (float)p.P, bh = rh / (float)p.P;
    const float cnt = (float)(p.sr * p.sr);

    for (int c0 = lane * 4; c0 < p.C; c0 += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < p.sr; ++iy) {
            float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)p.sr;
            for (int ix = 0; ix < p.sr; ++ix) {
                float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)p.sr;
                float yy = y;
                if (yy < -1.f || yy > (float)H || x < -1.f || x > (float)W) continue;
                yy = fmaxf(yy, 0.f);
                x = fmaxf(x, 0.f);
                int yl = (int)yy, xl = (int)x, yh, xh;
                if (yl >= H - 1) { yl = yh = H - 1; yy = (float)yl; } else { yh = yl + 1; }
                if (xl >= W - 1) { xl = xh = W - 1; x = (float)xl; } else { xh = xl + 1; }
                const float ly = yy - (float)yl, lx = x - (float)xl;
                const float hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const tv4 t1 = *reinterpret_cast<const tv4*>(f + ((size_t)yl * W + xl) * p.C + c0);
                const tv4 t2 = *reinterpret_cast<const tv4*>(f + ((size_t)yl * W + xh) * p.C + c0);
                const tv4 t3 = *reinterpret_cast<const tv4*>(f + ((size_t)yh * W + xl) * p.C + c0);
                const tv4 t4 = *reinterpret_cast<const tv4*>(f + ((size_t)yh * W + xh) * p.C + c0);
                const f32x4 v1 = __builtin_convertvector(t1, f32x4), v2 = __builtin_convertvector(t2, f32x4);
                const f32x4 v3 = __builtin_convertvector(t3, f32x4), v4 = __builtin_convertvector(t4, f32x4);
                acc = sample_acc(acc, w1, v1, w2, v2, w3, v3, w4, v4);
            }
        }
        acc /= cnt;
        *reinterpret_cast<tv4*>((T*)p.out + (size_t)bin * p.C + c0) = __builtin_convertvector(acc, tv4);
    }
}

constexpr int RA_WCAP = 40;      // feature pixels per staged row segment (40 KB of LDS per fp32 block: four blocks per CU)
constexpr int RA_CB = 64;        // channels per block

template <typename T>
__global__ __launch_bounds__(256) void roi_align_lds_kernel(const RoiArgs p) {
    typedef T tv4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) T rows[4][RA_WCAP][RA_CB];
    const int tid = threadIdx.x, cg = tid & 15, pw = tid >> 4;
    const int nq = p.C / RA_CB;
    const int k = blockIdx.x / nq, c0 = (blockIdx.x - k * nq) * RA_CB + cg * 4;

    const float* r = p.rois + (size_t)k * 5;
    const int bidx = (int)r[0];
    const float bx1 = r[1], by1 = r[2], bx2 = r[3], by2 = r[4];
    const int lvl = p.levels ? p.levels[k] : map_level(bx1, by1, bx2, by2, p.k_min);
    const int H = lvl == 0 ? p.h[0] : lvl == 1 ? p.h[1] : lvl == 2 ? p.h[2] : p.h[3];
    const int W = lvl == 0 ? p.w[0] : lvl == 1 ? p.w[1] : lvl == 2 ? p.w[2] : p.w[3];
    const float sc = lvl == 0 ? p.scale[0] : lvl == 1 ? p.scale[1] : lvl == 2 ? p.scale[2] : p.scale[3];
    const T* fb = (const T*)(lvl == 0 ? p.feat[0] : lvl == 1 ? p.feat[1] : lvl == 2 ? p.feat[2] : p.feat[3]);
    const T* f = fb + (size_t)bidx * H * W * p.C;

    const float x1 = bx1 * sc, y1 = by1 * sc, x2 = bx2 * sc, y2 = by2 * sc;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bw = rw / (float)p.P, bh = rh / (float)p.P;
    const float cnt = 4.f;

    // column range of the taps of all samples of a bin row (the same for every row); the sample abscissae are increasing
    auto tap_x = [&](float x, int& xl, int& xh, float& xx) {
        xx = fmaxf(x, 0.f);
        xl = (int)xx;
        if (xl >= W - 1) { xl = xh = W - 1; xx = (float)xl; } else { xh = xl + 1; }
    };
    int xlo, xhi;
    {
        float t;
        int a, b;
        tap_x(x1 + (float)0 * bw + ((float)0 + 0.5f) * bw / 2.f, xlo, a, t);
        tap_x(x1 + (float)(p.P - 1) * bw + ((float)1 + 0.5f) * bw / 2.f, b, xhi, t);
    }
    const int wpx = xhi - xlo + 1;
    const bool staged = wpx <= RA_WCAP;           // block-uniform
    const int per_row = wpx * (RA_CB / 4);        // 16-byte vectors per staged row segment
    const int total = 4 * per_row;

    // rows of bin row ph: (yl, yh) of its two sample rows, as the gather kernel computes them
    auto tap_y = [&](int ph, int iy, int& yl, int& yh, float& yy, bool& ok) {
        const float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / 2.f;
        ok = !(y < -1.f || y > (float)H);
        yy = fmaxf(y, 0.f);
        yl = (int)yy;
        if (yl >= H - 1) { yl = yh = H - 1; yy = (float)yl; } else { yh = yl + 1; }
    };
    constexpr int NST = 4 * RA_WCAP * (RA_CB / 4) / 256;      // staging vectors per thread
    tv4 st[NST];
    // which vector of which staged row each of this thread's staging slots moves: fixed for the whole ROI (only the row numbers
    // change from bin row to bin row), so the divisions are done once
    int g_off[NST], l_off[NST], s_row[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int v = min(tid + 256 * i, total - 1);   // slots past the segment re-read its last vector and are never stored
        const int rr = v / per_row, rem = v - rr * per_row, px = rem >> 4, c4 = rem & 15;
        s_row[i] = rr;
        g_off[i] = (xlo + px) * p.C + (c0 - cg * 4) + c4 * 4;
        l_off[i] = (tid + 256 * i < total) ? (rr * RA_WCAP + px) * RA_CB + c4 * 4 : -1;
    }
    auto prefetch = [&](int ph) {
        int yr[4];
        float t;
        bool ok;
        tap_y(ph, 0, yr[0], yr[1], t, ok);
        tap_y(ph, 1, yr[2], yr[3], t, ok);
#pragma unroll
        for (int i = 0; i < NST; ++i) {               // unconditional loads (a branch around a load serialises it behind vmcnt(0))
            const int row = s_row[i] == 0 ? yr[0] : s_row[i] == 1 ? yr[1] : s_row[i] == 2 ? yr[2] : yr[3];
            st[i] = *reinterpret_cast<const tv4*>(f + (size_t)row * W * p.C + g_off[i]);
        }
    };
    auto commit = [&]() {
        T* base = &rows[0][0][0];
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if (l_off[i] >= 0) *reinterpret_cast<tv4*>(base + l_off[i]) = st[i];
    };

    if (staged) prefetch(0);
    for (int ph = 0; ph < p.P; ++ph) {
        if (staged) {
            __syncthreads();                       // the previous bin row's reads of `rows` are done
            commit();
            __syncthreads();
            if (ph + 1 < p.P) prefetch(ph + 1);    // in flight under this bin row's arithmetic
        }
        if (pw < p.P) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int iy = 0; iy < 2; ++iy) {
                int yl, yh;
                float yy;
                bool yok;
                tap_y(ph, iy, yl, yh, yy, yok);
#pragma unroll
                for (int ix = 0; ix < 2; ++ix) {
                    float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / 2.f;
                    if (!yok || x < -1.f || x > (float)W) continue;
                    int xl, xh;
                    tap_x(x, xl, xh, x);
                    const float ly = yy - (float)yl, lx = x - (float)xl;
                    const float hy = 1.f - ly, hx = 1.f - lx;
                    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                    tv4 t1, t2, t3, t4;
                    if (staged) {
                        t1 = *reinterpret_cast<const tv4*>(&rows[2 * iy][xl - xlo][cg * 4]);
                        t2 = *reinterpret_cast<const tv4*>(&rows[2 * iy][xh - xlo][cg * 4]);
                        t3 = *reinterpret_cast<const tv4*>(&rows[2 * iy + 1][xl - xlo][cg * 4]);
                        t4 = *reinterpret_cast<const tv4*>(&rows[2 * iy + 1][xh - xlo][cg * 4]);
                    } else {
                        t1 = *reinterpret_cast<const tv4*>(f + ((size_t)yl * W + xl) * p.C + c0);
                        t2 = *reinterpret_cast<const tv4*>(f + ((size_t)yl * W + xh) * p.C + c0);
                        t3 = *reinterpret_cast<const tv4*>(f + ((size_t)yh * W + xl) * p.C + c0);
                        t4 = *reinterpret_cast<const tv4*>(f + ((size_t)yh * W + xh) * p.C + c0);
                    }
                    const f32x4 v1 = __builtin_convertvector(t1, f32x4), v2 = __builtin_convertvector(t2, f32x4);
                    const f32x4 v3 = __builtin_convertvector(t3, f32x4), v4 = __builtin_convertvector(t4, f32x4);
                    acc = sample_acc(acc, w1, v1, w2, v2, w3, v3, w4, v4);
                }
            }
            acc /= cnt;
            *reinterpret_cast<tv4*>((T*)p.out + ((size_t)k * p.P * p.P + (size_t)ph * p.P + pw) * p.C + c0) = __builtin_convertvector(acc, tv4);
        }
    }
}

// ---- second staged form: one block per (ROI quadrant of bins, 64-channel quarter).  The quadrant's WHOLE footprint (<= 16 x 16
// feature pixels x 256 B = 64 KB) is fetched at once -- every thread's loads in flight together -- then ONE barrier, then the <= 49
// bins x 16 channel groups come from LDS.  No chain of dependent stages; a footprint that does not fit takes the gather path.
constexpr int RQ_PX = 16;        // footprint rows / columns that fit

template <typename T, int CB>
__global__ __launch_bounds__(256) void roi_align_quad_kernel(const RoiArgs p) {
    constexpr int LPC = CB / 4;                 // lanes per pixel (a lane owns 4 channels)
    constexpr int RPP = 256 / LPC / RQ_PX;      // footprint rows one staging pass of the block covers (1 at 64 channels, 2 at 32)
    constexpr int NST = RQ_PX / RPP;
    typedef T tv4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) T fp[RQ_PX][RQ_PX][CB];
    __shared__ __attribute__((aligned(16))) f32x4 s_w[256];          // per sample of the quadrant: the four bilinear weights ...
    __shared__ __attribute__((aligned(16))) i32x4 s_o[256];          // ... and the four tap offsets into fp (elements; < 0: sample outside)
    const int tid = threadIdx.x, cg = tid & (LPC - 1);
    const int nq = p.C / CB;
    const int quad = blockIdx.x & 3, rest = blockIdx.x >> 2;
    const int k = rest / nq, cq = rest - k * nq;
    const int hp = (p.P + 1) >> 1;                                   // bins per quadrant side (7 of 14; 4 + 3 of 7)
    const int ph0 = (quad >> 1) * hp, ph1 = min(p.P, ph0 + hp), pw0 = (quad & 1) * hp, pw1 = min(p.P, pw0 + hp);
    const int nbh = ph1 - ph0, nbw = pw1 - pw0;

    const float* r = p.rois + (size_t)k * 5;
    const int bidx = (int)r[0];
    const float bx1 = r[1], by1 = r[2], bx2 = r[3], by2 = r[4];
    const int lvl = p.levels ? p.levels[k] : map_level(bx1, by1, bx2, by2, p.k_min);
    const int H = lvl == 0 ? p.h[0] : lvl == 1 ? p.h[1] : lvl == 2 ? p.h[2] : p.h[3];
    const int W = lvl == 0 ? p.w[0] : lvl == 1 ? p.w[1] : lvl == 2 ? p.w[2] : p.w[3];
    const float sc = lvl == 0 ? p.scale[0] : lvl == 1 ? p.scale[1] : lvl == 2 ? p.scale[2] : p.scale[3];
    const T* fb = (const T*)(lvl == 0 ? p.feat[0] : lvl == 1 ? p.feat[1] : lvl == 2 ? p.feat[2] : p.feat[3]);
    const T* f = fb + (size_t)bidx * H * W * p.C + cq * CB;

    const float x1 = bx1 * sc, y1 = by1 * sc, x2 = bx2 * sc, y2 = by2 * sc;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bw = rw / (float)p.P, bh = rh / (float)p.P;

    auto tap = [&](float v, int lim, int& lo, int& hi, float& vv) {      // the gather kernel's clamping of one coordinate
        vv = fmaxf(v, 0.f);
        lo = (int)vv;
        if (lo >= lim - 1) { lo = hi = lim - 1; vv = (float)lo; } else { hi = lo + 1; }
    };
    auto sx = [&](int pw, int ix) { return x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / 2.f; };
    auto sy = [&](int ph, int iy) { return y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / 2.f; };
    int xlo, xhi, ylo, yhi;
    {
        float t;
        int a;
        tap(sx(pw0, 0), W, xlo, a, t);
        tap(sx(pw1 - 1, 1), W, a, xhi, t);
        tap(sy(ph0, 0), H, ylo, a, t);
        tap(sy(ph1 - 1, 1), H, a, yhi, t);
    }
    const int wpx = xhi - xlo + 1, hpx = yhi - ylo + 1;
    const bool staged = wpx <= RQ_PX && hpx <= RQ_PX;               // block-uniform
    if (staged) {
        // thread -> (row phase, pixel column, channel group): one 16-byte vector per RPP footprint rows, all requested before any is stored
        const int px = (tid / LPC) % RQ_PX, pr = tid / (LPC * RQ_PX);
        tv4 st[NST];
        const bool col_ok = px < wpx;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int py = i * RPP + pr;
            const int yy = min(ylo + py, yhi), xx = min(xlo + px, xhi);
            // rows and columns past the footprint issue no memory request
            if (py < hpx && col_ok) st[i] = *reinterpret_cast<const tv4*>(f + ((size_t)yy * W + xx) * p.C + cg * 4);
        }
        // while the loads fly: the bilinear taps of every sample of the quadrant, ONCE per block (196 records for 7 x 7 bins) instead of
        // once per channel-group lane -- the coordinate arithmetic is what bounds this operator (about 40 VALU instructions per sample,
        // repeated by the 64 channel lanes of a bin in the gather kernel)
        if (tid < 4 * nbh * nbw) {
            const int bin = tid >> 2, iy = (tid >> 1) & 1, ix = tid & 1;
            const int bh_i = bin / nbw, ph = ph0 + bh_i, pw = pw0 + (bin - bh_i * nbw);
            const float y = sy(ph, iy);
            float x = sx(pw, ix);
            const bool ok = !(y < -1.f || y > (float)H || x < -1.f || x > (float)W);
            int yl, yh, xl, xh;
            float yy;
            tap(y, H, yl, yh, yy);
            tap(x, W, xl, xh, x);
            const float ly = yy - (float)yl, lx = x - (float)xl;
            const float hy = 1.f - ly, hx = 1.f - lx;
            s_w[tid] = f32x4{hy * hx, hy * lx, ly * hx, ly * lx};
            const int r0 = (yl - ylo) * RQ_PX, r1 = (yh - ylo) * RQ_PX, c0 = xl - xlo, c1 = xh - xlo;
            s_o[tid] = ok ? i32x4{(r0 + c0) * CB, (r0 + c1) * CB, (r1 + c0) * CB, (r1 + c1) * CB} : i32x4{-1, -1, -1, -1};
        }
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if (i * RPP + pr < hpx && col_ok) *reinterpret_cast<tv4*>(&fp[i * RPP + pr][px][cg * 4]) = st[i];
        __syncthreads();
        const T* base = &fp[0][0][cg * 4];
        for (int t = tid / LPC; t < nbh * nbw; t += 256 / LPC) {
            const int bh_i = t / nbw, ph = ph0 + bh_i, pw = pw0 + (t - bh_i * nbw);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int smp = 0; smp < 4; ++smp) {                     // (iy, ix) = (0,0) (0,1) (1,0) (1,1): the gather kernel's order
                const i32x4 o = s_o[4 * t + smp];
                if (o[0] < 0) continue;
                const f32x4 w = s_w[4 * t + smp];
                const f32x4 v1 = __builtin_convertvector(*reinterpret_cast<const tv4*>(base + o[0]), f32x4);
                const f32x4 v2 = __builtin_convertvector(*reinterpret_cast<const tv4*>(base + o[1]), f32x4);
                const f32x4 v3 = __builtin_convertvector(*reinterpret_cast<const tv4*>(base + o[2]), f32x4);
                const f32x4 v4 = __builtin_convertvector(*reinterpret_cast<const tv4*>(base + o[3]), f32x4);
                acc = sample_acc(acc, w[0], v1, w[1], v2, w[2], v3, w[3], v4);
            }
            acc /= 4.f;
            *reinterpret_cast<tv4*>((T*)p.out + ((size_t)k * p.P * p.P + (size_t)ph * p.P + pw) * p.C + cq * CB + cg * 4) =
                __builtin_convertvector(acc, tv4);
        }
        return;
    }
    // footprint larger than the LDS window: every lane gathers its taps from L1 / L2, as roi_align_kernel does
    for (int t = tid / LPC; t < nbh * nbw; t += 256 / LPC) {
        const int bh_i = t / nbw, ph = ph0 + bh_i, pw = pw0 + (t - bh_i * nbw);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int iy = 0; iy < 2; ++iy) {
            const float y = sy(ph, iy);
            const bool yok = !(y < -1.f || y > (float)H);
            int yl, yh;
            float yy;
            tap(y, H, yl, yh, yy);
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                float x = sx(pw, ix);
                if (!yok || x < -1.f || x > (float)W) continue;
                int xl, xh;
                tap(x, W, xl, xh, x);
                const float ly = yy - (float)yl, lx = x - (float)xl;
                const float hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const tv4 t1 = *reinterpret_cast<const tv4*>(f + ((size_t)yl * W + xl) * p.C + cg * 4);
                const tv4 t2 = *reinterpret_cast<const tv4*>(f + ((size_t)yl * W + xh) * p.C + cg * 4);
                const tv4 t3 = *reinterpret_cast<const tv4*>(f + ((size_t)yh * W + xl) * p.C + cg * 4);
                const tv4 t4 = *reinterpret_cast<const tv4*>(f + ((size_t)yh * W + xh) * p.C + cg * 4);
                const f32x4 v1 = __builtin_convertvector(t1, f32x4), v2 = __builtin_convertvector(t2, f32x4);
                const f32x4 v3 = __builtin_convertvector(t3, f32x4), v4 = __builtin_convertvector(t4, f32x4);
                acc = sample_acc(acc, w1, v1, w2, v2, w3, v3, w4, v4);
            }
        }
        acc /= 4.f;
        *reinterpret_cast<tv4*>((T*)p.out + ((size_t)k * p.P * p.P + (size_t)ph * p.P + pw) * p.C + cq * CB + cg * 4) =
            __builtin_convertvector(acc, tv4);
    }
}


}  // namespace

extern "C" void seam_roi_align_set_lds(int on) { seam_opt::g_value[seam_opt::ROIALIGN_LDS].store(on < 0 ? 0 : on > 2 ? 2 : on, std::memory_order_relaxed); }

template <typename T>
static int roi_align_launch(const void* feat0, const void* feat1, const void* feat2, const void* feat3, const int* hw, int C,
                            float scale0, float scale1, float scale2, float scale3, int k_min, const float* rois,
                            const int* levels, void* out, int K, int P, int sampling_ratio, void* stream) {
    if (K <= 0) return 0;
    if ((C & 3) || C > 1024 * 4) return (int)hipErrorInvalidValue;
    RoiArgs a;
    a.feat[0] = feat0; a.feat[1] = feat1; a.feat[2] = feat2; a.feat[3] = feat3;
    for (int i = 0; i < 4; ++i) { a.h[i] = hw[2 * i]; a.w[i] = hw[2 * i + 1]; }
    a.scale[0] = scale0; a.scale[1] = scale1; a.scale[2] = scale2; a.scale[3] = scale3;
    a.C = C; a.k_min = k_min; a.rois = rois; a.levels = levels; a.out = out; a.K = K; a.P = P;
    a.sr = sampling_ratio;
    const int g_roi_lds = seam_opt::get(seam_opt::ROIALIGN_LDS);      // 0 gather, 1 row-staged tiles, 2 (default) quadrant tiles
    if (g_roi_lds == 2 && sampling_ratio == 2 && P <= 16 && (C % RA_CB) == 0) {
        // (64 channels per block; 32 -- more blocks per CU -- measured 2.2x slower in fp32: 128-byte pixel rows alias in LDS)
        hipLaunchKernelGGL((roi_align_quad_kernel<T, RA_CB>), dim3((unsigned)((long)K * (C / RA_CB) * 4)), dim3(256), 0, (hipStream_t)stream, a);
        return (int)hipGetLastError();
    }
    if (g_roi_lds == 1 && sampling_ratio == 2 && P <= 16 && (C % RA_CB) == 0) {
        hipLaunchKernelGGL(roi_align_lds_kernel<T>, dim3((unsigned)((long)K * (C / RA_CB))), dim3(256), 0, (hipStream_t)stream, a);
        return (int)hipGetLastError();
    }
    const long bins = (long)K * P * P;
    hipLaunchKernelGGL(roi_align_kernel<T>, dim3((unsigned)((bins + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int seam_roi_align_f32(const float* feat0, const float* feat1, const float* feat2, const float* feat3,
                                  const int* hw, int C, float scale0, float scale1, float scale2, float scale3,
                                  int k_min, const float* rois, const int* levels, float* out, int K, int P,
                                  int sampling_ratio, void* stream) {
    return roi_align_launch<float>(feat0, feat1, feat2, feat3, hw, C, scale0, scale1, scale2, scale3, k_min, rois, levels,
                                   out, K, P, sampling_ratio, stream);
}

extern "C" int seam_roi_align_f16(const void* feat0, const void* feat1, const void* feat2, const void* feat3,
                                  const int* hw, int C, float scale0, float scale1, float scale2, float scale3,
                                  int k_min, const float* rois, const int* levels, void* out, int K, int P,
                                  int sampling_ratio, void* stream) {
    return roi_align_launch<_Float16>(feat0, feat1, feat2, feat3, hw, C, scale0, scale1, scale2, scale3, k_min, rois,
                                      levels, out, K, P, sampling_ratio, stream);
}
