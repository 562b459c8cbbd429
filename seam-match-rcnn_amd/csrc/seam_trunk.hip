// seam_trunk.hip -- seam_match_trunk_f32: the match trunk of MatchPredictor / TemporalAggregationNLB as ONE ABI call
// (ref models/match_head.py:50-62,67-69,93-95: conv_seq = 4 valid 3x3 convs + ReLU, AvgPool2d(6,6) + ReLU,
// Linear(1024,256) + BatchNorm1d).
//
// One call, six launches.  The kernels stay separate on purpose (measured, profiles/r03a_bench_kernel_stats.csv):
//   * avg-pool + Linear are 107.8 us + 29.4 us per trunk = 0.27 ms of the 120 ms step (0.23 %);
//   * pooling inside the last conv's epilogue needs a reduction over the 9 Winograd tiles of a ROI, which straddle the 32-tile
//     blocks (3.55 ROIs per block): either float atomics -- the sums would no longer be bit-reproducible run to run, which the
//     parity tests assert -- or ROI-aligned tile groups of 27 that leave 15.6 % of the MFMA slots of a 3.9 ms layer empty
//     (+0.6 ms, more than the fusion saves);
//   * the Linear needs all 1024 pooled channels of a ROI, i.e. a cross-block reduction over the 32 channel tiles of that conv: a
//     second kernel either way.
// The form of each conv (implicit GEMM / Winograd F(2x2) / F(2x4)) is chosen exactly as the Python layer chooses it: on the map
// geometry alone, so that a ROI's descriptor does not depend on the batch it rides in.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "seam_hip.h"

extern "C" {

int64_t seam_match_trunk_workspace_floats(int K) {
    return (int64_t)K * (12 * 12 * 256 + 10 * 10 * 256 + 8 * 8 * 256 + 6 * 6 * 1024 + 1024) + 64;
}

// form of one stride-1 valid 3x3 layer on an h x h map: 0 implicit GEMM, 1 F(2x2,3x3), 2 F(2x4,3x3)  (ops.conv2d's rule)
static int trunk_form(const seam_trunk_layer_t& l, int h, int c, int k) {
    if (!l.u || !seam_wino_supported(c, k, 3, 3, 1)) return 0;
    if (seam_wino_slot_fill_pct(256, h, h, c, k, 0) < 55) return 0;
    if (l.u24) {
        const long long s24 = seam_wino24_issue_slots(256, h, h, c, k, 0), s22 = seam_wino_issue_slots(256, h, h, c, k, 0);
        if (s24 > 0 && (double)s24 <= 0.95 * (double)s22) return 2;
    }
    return 1;
}

int seam_match_trunk_f32(const float* roi, const seam_trunk_layer_t* conv, const seam_trunk_layer_t* linear, float* x3, int K,
                         float* ws, const int* form, seam_stream_t stream) {
    if (K <= 0) return 0;
    if (!roi || !conv || !linear || !x3 || !ws) return (int)hipErrorInvalidValue;
    static const int cin[4] = {256, 256, 256, 256}, cout[4] = {256, 256, 256, 1024}, hin[4] = {14, 12, 10, 8};
    float* buf[5];
    float* q = ws;
    for (int i = 0; i < 4; ++i) {
        buf[i] = q;
        q += (size_t)K * (hin[i] - 2) * (hin[i] - 2) * cout[i];
    }
    buf[4] = q;                                     // pooled [K,1024]
    const float* x = roi;
    for (int i = 0; i < 4; ++i) {
        const seam_trunk_layer_t& l = conv[i];
        const int f = form ? form[i] : trunk_form(l, hin[i], cin[i], cout[i]);
        int rc;
        if (f == 2 && l.u24)
            rc = seam_conv3x3_wino24_f32(x, l.u24, l.scale, l.shift, nullptr, buf[i], K, hin[i], hin[i], cin[i], cout[i], 0, 1, stream);
        else if (f == 1 && l.u)
            rc = seam_conv3x3_wino_f32(x, l.u, l.scale, l.shift, nullptr, buf[i], K, hin[i], hin[i], cin[i], cout[i], 0, 1, stream);
        else if (l.w)
            rc = seam_conv2d_f32(x, l.w, l.scale, l.shift, nullptr, buf[i], K, hin[i], hin[i], cin[i], cout[i], 3, 3, 1, 0, 1, stream);
        else
            rc = (int)hipErrorInvalidValue;
        if (rc) return rc;
        x = buf[i];
    }
    int rc = seam_avgpool_f32(buf[3], buf[4], K, 36, 1024, stream);      // AvgPool2d(6,6); its ReLU is the identity on means of ReLUs
    if (rc) return rc;
    return seam_conv2d_f32(buf[4], linear->w, linear->scale, linear->shift, nullptr, x3, K, 1, 1, 1024, 256, 1, 1, 1, 0, 0, stream);
}

}  // extern "C"
