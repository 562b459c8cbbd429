// seam_detect.hip -- detection post-processing kernels (gfx950): BoxCoder.decode + clip, greedy NMS
// (64x64 bitmask tiles + one-wave scan), mask-channel select.  Latency-bound integer/bit work.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr float kXformClip = 4.135166556742356f;   // log(1000/16)

__global__ void decode_boxes_kernel(const float* __restrict__ deltas, const float* __restrict__ boxes,
                                    float* __restrict__ out, int N, int ncls, float wx, float wy, float ww, float wh,
                                    float clip_h, float clip_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * ncls) return;
    const int n = i / ncls;
    const float4 bx = reinterpret_cast<const float4*>(boxes)[n];
    const float4 d = reinterpret_cast<const float4*>(deltas)[i];
    const float w = bx.z - bx.x, h = bx.w - bx.y;
    const float cx = bx.x + 0.5f * w, cy = bx.y + 0.5f * h;
    const float dx = d.x / wx, dy = d.y / wy;
    const float dw = fminf(d.z / ww, kXformClip), dh = fminf(d.w / wh, kXformClip);
    const float pcx = dx * w + cx, pcy = dy * h + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
    if (clip_w > 0.f) {
        x1 = fminf(fmaxf(x1, 0.f), clip_w); x2 = fminf(fmaxf(x2, 0.f), clip_w);
        y1 = fminf(fmaxf(y1, 0.f), clip_h); y2 = fminf(fmaxf(y2, 0.f), clip_h);
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(x1, y1, x2, y2);
}

// mask[i][jb] bit j: box (jb*64+j) overlaps box i with IoU > thr (only j > i matters)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, uint64_t* __restrict__ mask,
                                                      int N, float thr) {
    const int ib = blockIdx.y, jb = blockIdx.x;
    if (jb < ib) return;
    boxes += (size_t)blockIdx.z * N * 4;                 // batch (image) index
    mask += (size_t)blockIdx.z * N * gridDim.x;
    __shared__ float4 cb[64];
    const int t = threadIdx.x;
    const int j = jb * 64 + t;
    if (j < N) cb[t] = reinterpret_cast<const float4*>(boxes)[j];
    __syncthreads();
    const int i = ib * 64 + t;
    if (i >= N) return;
    const float4 a = reinterpret_cast<const float4*>(boxes)[i];
    const float aa = (a.z - a.x) * (a.w - a.y);
    const int nj = min(64, N - jb * 64);
    uint64_t bits = 0;
    for (int k = (ib == jb ? t + 1 : 0); k < nj; ++k) {
        const float4 c = cb[k];
        const float w = fmaxf(fminf(a.z, c.z) - fmaxf(a.x, c.x), 0.f);
        const float h = fmaxf(fminf(a.w, c.w) - fmaxf(a.y, c.y), 0.f);
        const float inter = w * h;
        const float iou = inter / (aa + (c.z - c.x) * (c.w - c.y) - inter);
        if (iou > thr) bits |= 1ull << k;
    }
    mask[(size_t)i * gridDim.x + jb] = bits;
}

// one wave per image: walk the 64-box blocks in order; resolve each diagonal block with scalar bit ops,
// then OR the kept rows into the running "removed" words (lane l owns words l, l+64, l+128, l+192 of
// <= 256 words, i.e. N <= 16384).  The rows of a block are fetched with UNCONDITIONAL loads in groups of 8
// (independent addresses, all in flight together) and masked by the kept bit afterwards -- a load per kept
// box behind its own wait costs one memory latency per survivor.  max_keep > 0: stop after that many
// survivors (greedy NMS decides box i from higher-scored boxes only, so the first max_keep survivors of
// the full scan ARE the scan's result truncated to max_keep); everything behind gets keep = 0.
__global__ __launch_bounds__(64) void nms_scan_kernel(const uint64_t* __restrict__ mask, int* __restrict__ keep,
                                                      int N, int nb, int max_keep) {
    const int lane = threadIdx.x;
    mask += (size_t)blockIdx.x * N * nb;
    keep += (size_t)blockIdx.x * N;
    uint64_t rem0 = 0, rem1 = 0, rem2 = 0, rem3 = 0;   // removed bits of words lane, lane+64, lane+128, lane+192
    int total = 0;
    int blk = 0;
    for (; blk < nb; ++blk) {
        // removed word of this block lives in lane (blk & 63), slot (blk >> 6)
        const int slot = blk >> 6;
        const uint64_t mine = slot == 0 ? rem0 : slot == 1 ? rem1 : slot == 2 ? rem2 : rem3;
        uint64_t removed = __shfl(mine, blk & 63, 64);
        const int i = blk * 64 + lane;
        const uint64_t diag = i < N ? mask[(size_t)i * nb + blk] : 0ull;
        const int nvalid = min(64, N - blk * 64);
        uint64_t kept = 0;
        for (int b = 0; b < nvalid; ++b) {
            const uint64_t row = __shfl(diag, b, 64);     // wave-uniform
            if (!((removed >> b) & 1ull)) {
                kept |= 1ull << b;
                removed |= row;
            }
        }
        if (max_keep > 0) {                               // truncate to the first (max_keep - total) survivors of this block
            const int room = max_keep - total;
            int c = __popcll((unsigned long long)kept);
            while (c > room) {                            // drop the highest set bits (lowest scores)
                kept &= ~(1ull << (63 - __clzll((unsigned long long)kept)));
                --c;
            }
            total += c;
        }
        if (i < N) keep[i] = (int)((kept >> lane) & 1ull);
        if (max_keep > 0 && total >= max_keep) { ++blk; break; }
        // propagate kept rows to later words
        const bool u0 = lane > blk && lane < nb, u1 = lane + 64 > blk && lane + 64 < nb;
        const bool u2 = lane + 128 > blk && lane + 128 < nb, u3 = lane + 192 > blk && lane + 192 < nb;
        const uint64_t* base = mask + (size_t)(blk * 64) * nb;
        for (int b0 = 0; b0 < nvalid; b0 += 8) {
            if (!((kept >> b0) & 0xFFull)) continue;      // wave-uniform
            uint64_t v0[8], v1[8], v2[8], v3[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int b = min(b0 + e, nvalid - 1);
                const uint64_t* r = base + (size_t)b * nb;
                v0[e] = u0 ? r[lane] : 0ull;
                v1[e] = u1 ? r[lane + 64] : 0ull;
                v2[e] = u2 ? r[lane + 128] : 0ull;
                v3[e] = u3 ? r[lane + 192] : 0ull;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint64_t on = (b0 + e < nvalid && ((kept >> (b0 + e)) & 1ull)) ? ~0ull : 0ull;
                rem0 |= v0[e] & on; rem1 |= v1[e] & on; rem2 |= v2[e] & on; rem3 |= v3[e] & on;
            }
        }
    }
    for (int i = blk * 64 + lane; i < N; i += 64) keep[i] = 0;      // behind the early exit
}

// ------------------------------------------------------------------------------------------------
// RPN filter_proposals, first half [TV RegionProposalNetwork.filter_proposals / _get_top_n_idx], for ONE pyramid level of a
// batch: exact top-k of the n = H*W*A objectness logits of each image (order: logit descending, anchor index ascending --
// a stable descending sort's first k), then for the k winners in that order: gather the 4 deltas and the anchor, BoxCoder.decode
// (weights 1,1,1,1, dw/dh clamped to log(1000/16)), clip to the image, sigmoid of the logit.  One 1024-thread workgroup per
// image: MSB-first 8-bit radix select of the k-th largest key (4 LDS histogram passes over the logits, L2-resident), one
// collection pass, rank counting over the k winners in LDS.  Replaces a full device sort of every level (120 000 keys per
// image at 200x200x3 to keep 1000).  Logits / deltas are read in place from the fused head output [N,H,W,A+4A] (strided).
constexpr int RPN_TK_MAX = 1024;

__device__ __forceinline__ unsigned rpn_key(float x) {
    if (x != x) x = -INFINITY;                   // NaN ranks last
    x += 0.f;                                    // -0 -> +0
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct RpnTkArgs {
    const float* obj; const float* dlt; const float* anchors; const float* clip_hw;
    float* boxes; float* scores; int64_t* index;
    int64_t obj_img_stride, dlt_img_stride;      // floats between images
    int obj_pix_stride, dlt_pix_stride, A, n, k;
    int64_t out_img_stride;                      // rows (candidates) per image in boxes / scores / index
    int out_offset;                              // first row of this level inside an image's rows
};

__global__ __launch_bounds__(1024) void rpn_topk_decode_kernel(const RpnTkArgs p) {
    __shared__ unsigned hist[256];
    __shared__ unsigned skey[RPN_TK_MAX];
    __shared__ int sidx[RPN_TK_MAX];
    __shared__ unsigned s_prefix, s_krem, s_cnt, s_neq, s_tie;
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, img = blockIdx.x, A = p.A, n = p.n, k = p.k;
    const float* obj = p.obj + (size_t)img * p.obj_img_stride;
    auto logit = [&](int j) -> float { const int px = j / A; return obj[(size_t)px * p.obj_pix_stride + (j - px * A)]; };
    unsigned prefix = 0, msk = 0, krem = (unsigned)k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int j = tid; j < n; j += 1024) {
            const unsigned key = rpn_key(logit(j));
            if ((key & msk) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0;
            int bsel = 0;
            for (int bb = 255; bb >= 0; --bb) {
                if (cum + hist[bb] >= krem) { bsel = bb; break; }
                cum += hist[bb];
            }
            s_prefix = prefix | ((unsigned)bsel << shift);
            s_krem = krem - cum;
            s_neq = hist[bsel];                  // after the last pass: number of items whose key == T
        }
        __syncthreads();
        prefix = s_prefix;
        krem = s_krem;
        msk |= 0xFFu << shift;
    }
    const unsigned T = prefix;                   // k-th largest key; krem (>= 1) of the s_neq items equal to T are taken
    const unsigned nabove = (unsigned)k - krem;
    const bool all_ties = s_neq == krem;         // common case: every item equal to T is a winner
    if (tid == 0) { s_cnt = 0; s_tie = 0; }
    __syncthreads();
    for (int j = tid; j < n; j += 1024) {
        const unsigned key = rpn_key(logit(j));
        if (key > T || (all_ties && key == T)) {
            const unsigned pos = atomicAdd(&s_cnt, 1u);
            skey[pos] = key; sidx[pos] = j;
        }
    }
    __syncthreads();
    if (!all_ties) {
        // more items tie at T than there is room for: take them lowest anchor index first, walking the row in index order
        // (1024 consecutive indices per round, positions from a block-wide exclusive count)
        const int lane = tid & 63, wid = tid >> 6;
        for (int base = 0; base < n; base += 1024) {
            const unsigned got = s_tie;
            if (got >= krem) break;
            const int j = base + tid;
            const bool eq = j < n && rpn_key(logit(j)) == T;
            const unsigned long long bal = __ballot(eq);
            if (lane == 0) wsum[wid] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned before = 0, tot = 0;
            for (int w2 = 0; w2 < 16; ++w2) { const unsigned c = wsum[w2]; if (w2 < wid) before += c; tot += c; }
            const unsigned mypos = got + before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            if (eq && mypos < krem) { skey[nabove + mypos] = T; sidx[nabove + mypos] = j; }
            __syncthreads();
            if (tid == 0) s_tie = got + tot;
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid < k) {                               // order the k winners by counting; the reads of (skey[j], sidx[j]) are broadcasts
        const unsigned mk = skey[tid];
        const int mj = sidx[tid];
        int rank = 0;
        for (int j = 0; j < k; ++j) {
            const unsigned kj = skey[j];
            rank += (kj > mk || (kj == mk && sidx[j] < mj)) ? 1 : 0;
        }
        const int px = mj / A, a = mj - px * A;
        const float x = obj[(size_t)px * p.obj_pix_stride + a];
        const float* dp = p.dlt + (size_t)img * p.dlt_img_stride + (size_t)px * p.dlt_pix_stride + a * 4;   // 4-byte aligned only
        const float4 d = make_float4(dp[0], dp[1], dp[2], dp[3]);
        const float4 bx = reinterpret_cast<const float4*>(p.anchors)[mj];
        const float clip_h = p.clip_hw[img * 2], clip_w = p.clip_hw[img * 2 + 1];
        // same operation order as decode_boxes_kernel with weights (1,1,1,1)
        const float w = bx.z - bx.x, h = bx.w - bx.y;
        const float cx = bx.x + 0.5f * w, cy = bx.y + 0.5f * h;
        const float dw = fminf(d.z, kXformClip), dh = fminf(d.w, kXformClip);
        const float pcx = d.x * w + cx, pcy = d.y * h + cy;
        const float pw = expf(dw) * w, ph = expf(dh) * h;
        float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
        x1 = fminf(fmaxf(x1, 0.f), clip_w); x2 = fminf(fmaxf(x2, 0.f), clip_w);
        y1 = fminf(fmaxf(y1, 0.f), clip_h); y2 = fminf(fmaxf(y2, 0.f), clip_h);
        const size_t row = (size_t)img * p.out_img_stride + p.out_offset + rank;
        reinterpret_cast<float4*>(p.boxes)[row] = make_float4(x1, y1, x2, y2);
        p.scores[row] = 1.f / (1.f + expf(-x));
        if (p.index) p.index[row] = (int64_t)mj;
    }
}

// logits [K,14,14,4,ncls] (sub-pixel (a,b) groups) -> prob [K,1,28,28], channel labels[k]
template <typename T>
__global__ void mask_select_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                   float* __restrict__ prob, int K, int ncls) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * 784) return;
    const int k = i / 784;
    const int r = i - k * 784;
    const int y = r / 28, x = r - y * 28;
    const int h = y >> 1, a = y & 1, w = x >> 1, b = x & 1;
    const int lab = (int)labels[k];
    const float v = (float)logits[((((size_t)k * 14 + h) * 14 + w) * 4 + (a * 2 + b)) * ncls + lab];
    prob[i] = 1.f / (1.f + expf(-v));
}

// paste_masks_in_image [TV]: mask prob [K,1,28,28] zero-padded to 30x30, box expanded by 30/28 and
// truncated to int, bilinear (align_corners=False) resize of the padded map to the integer box size,
// pasted into [K,1,H,W] (zeros elsewhere).  One thread per output pixel quad.
__global__ void paste_masks_kernel(const float* __restrict__ masks, const float* __restrict__ boxes,
                                   float* __restrict__ out, int K, int H, int W) {
    const int k = blockIdx.y;
    const float4 bx = reinterpret_cast<const float4*>(boxes)[k];
    const float scale = 30.f / 28.f;
    const float wh = (bx.z - bx.x) * 0.5f * scale, hh = (bx.w - bx.y) * 0.5f * scale;
    const float xc = (bx.z + bx.x) * 0.5f, yc = (bx.w + bx.y) * 0.5f;
    const int x0 = (int)(xc - wh), y0 = (int)(yc - hh), x1 = (int)(xc + wh), y1 = (int)(yc + hh);   // trunc toward 0 (int64 cast)
    const int bw = max(x1 - x0 + 1, 1), bh = max(y1 - y0 + 1, 1);
    const float sx = 30.f / (float)bw, sy = 30.f / (float)bh;
    const float* m = masks + (size_t)k * 784;
    float* o = out + (size_t)k * H * W;
    const int total = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        float v = 0.f;
        const int ry = y - y0, rx = x - x0;
        if (ry >= 0 && ry < bh && rx >= 0 && rx < bw && y <= y1 && x <= x1) {
            float fy = sy * ((float)ry + 0.5f) - 0.5f, fx = sx * ((float)rx + 0.5f) - 0.5f;
            if (fy < 0.f) fy = 0.f;
            if (fx < 0.f) fx = 0.f;
            int iy = min((int)fy, 29), ix = min((int)fx, 29);
            const int iy1 = iy < 29 ? iy + 1 : iy, ix1 = ix < 29 ? ix + 1 : ix;
            const float ly = fminf(fmaxf(fy - (float)iy, 0.f), 1.f), lx = fminf(fmaxf(fx - (float)ix, 0.f), 1.f);
            auto at = [&](int yy, int xx) -> float {      // 30x30 zero-padded view of the 28x28 map
                return (yy >= 1 && yy <= 28 && xx >= 1 && xx <= 28) ? m[(yy - 1) * 28 + (xx - 1)] : 0.f;
            };
            v = (1.f - ly) * ((1.f - lx) * at(iy, ix) + lx * at(iy, ix1)) + ly * ((1.f - lx) * at(iy1, ix) + lx * at(iy1, ix1));
        }
        o[i] = v;
    }
}

// torchvision.ops.box_iou (ref evaluate_movingfashion.py:207): iou[i,j] of xyxy boxes a[i], b[j]
__global__ void box_iou_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int Na, int Nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Na * Nb) return;
    const float4 p = reinterpret_cast<const float4*>(a)[i / Nb];
    const float4 q = reinterpret_cast<const float4*>(b)[i % Nb];
    const float ap = (p.z - p.x) * (p.w - p.y), aq = (q.z - q.x) * (q.w - q.y);
    const float iw = fmaxf(fminf(p.z, q.z) - fmaxf(p.x, q.x), 0.f);
    const float ih = fmaxf(fminf(p.w, q.w) - fmaxf(p.y, q.y), 0.f);
    const float inter = iw * ih;
    out[i] = inter / (ap + aq - inter);
}

}  // namespace

extern "C" {

int seam_decode_boxes_f32(const float* deltas, const float* boxes_in, float* boxes_out, int N, int ncls, float wx,
                          float wy, float ww, float wh, float clip_h, float clip_w, void* stream) {
    if (N <= 0) return 0;
    const int total = N * ncls;
    hipLaunchKernelGGL(decode_boxes_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, deltas,
                       boxes_in, boxes_out, N, ncls, wx, wy, ww, wh, clip_h, clip_w);
    return (int)hipGetLastError();
}

int seam_nms_sorted_topn_f32(const float* boxes, int* keep, int B, int N, float thr, int max_keep, uint64_t* mask_ws, void* stream) {
    if (N <= 0 || B <= 0) return 0;
    if (N > 16384 || B > 65535 || max_keep < 0) return (int)hipErrorInvalidValue;
    const int nb = (N + 63) / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb, B), dim3(64), 0, (hipStream_t)stream, boxes, mask_ws, N, thr);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, mask_ws, keep, N, nb, max_keep);
    return (int)hipGetLastError();
}

int seam_nms_sorted_f32(const float* boxes, int* keep, int B, int N, float thr, uint64_t* mask_ws, void* stream) {
    return seam_nms_sorted_topn_f32(boxes, keep, B, N, thr, 0, mask_ws, stream);
}

int seam_rpn_topk_max(void) { return RPN_TK_MAX; }

int seam_rpn_topk_decode_f32(const float* obj, const float* deltas, const float* anchors, const float* clip_hw, float* boxes,
                             float* scores, int64_t* index, int n_img, int n, int A, int k, int64_t obj_img_stride,
                             int obj_pix_stride, int64_t dlt_img_stride, int dlt_pix_stride, int64_t out_img_stride,
                             int out_offset, void* stream) {
    if (n_img <= 0 || n <= 0 || k <= 0) return 0;
    if (k > RPN_TK_MAX || k > n || A <= 0 || n % A || n_img > 65535 || ((uintptr_t)anchors & 15) || ((uintptr_t)boxes & 15))
        return (int)hipErrorInvalidValue;
    RpnTkArgs p{obj, deltas, anchors, clip_hw, boxes, scores, index, obj_img_stride, dlt_img_stride, obj_pix_stride,
                dlt_pix_stride, A, n, k, out_img_stride, out_offset};
    hipLaunchKernelGGL(rpn_topk_decode_kernel, dim3(n_img), dim3(1024), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

int seam_paste_masks_f32(const float* masks, const float* boxes, float* out, int K, int H, int W, void* stream) {
    if (K <= 0) return 0;
    int gx = (H * W + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(paste_masks_kernel, dim3(gx, K), dim3(256), 0, (hipStream_t)stream, masks, boxes, out, K, H, W);
    return (int)hipGetLastError();
}

int seam_box_iou_f32(const float* a, const float* b, float* out, int Na, int Nb, void* stream) {
    if (Na <= 0 || Nb <= 0) return 0;
    hipLaunchKernelGGL(box_iou_kernel, dim3((Na * Nb + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, out, Na, Nb);
    return (int)hipGetLastError();
}

int seam_mask_select_f32(const float* logits, const int64_t* labels, float* prob, int K, int ncls, void* stream) {
    if (K <= 0) return 0;
    const int total = K * 784;
    hipLaunchKernelGGL(mask_select_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits,
                       labels, prob, K, ncls);
    return (int)hipGetLastError();
}

int seam_mask_select_f16(const void* logits, const int64_t* labels, float* prob, int K, int ncls, void* stream) {
    if (K <= 0) return 0;
    const int total = K * 784;
    hipLaunchKernelGGL(mask_select_kernel<_Float16>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)logits, labels, prob, K, ncls);
    return (int)hipGetLastError();
}

}  // extern "C"
