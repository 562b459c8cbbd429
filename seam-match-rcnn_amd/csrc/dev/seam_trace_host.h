// dev/seam_trace_host.h -- host side of the s_memtime stamp builds (-DSEAM_DEV_BUILD -DSEAM_*_TRACE=<block>; tools/experiments/*.sh).
// NOT part of the shipped library: the Makefile never defines SEAM_DEV_BUILD, and without it no launcher includes this file -- the
// C ABI's "never allocates, never synchronises" contract holds for every shipped entry point.
#pragma once
#ifndef SEAM_DEV_BUILD
#error "dev/seam_trace_host.h is for -DSEAM_DEV_BUILD experiment builds only"
#endif
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

namespace seam_dev {

struct TraceBuf {
    unsigned long long* dev = nullptr;
    size_t words = 0;
    int launches = 0;
};

// (re)zeroed device buffer of `words` stamps; allocated on first use
inline unsigned long long* trace_begin(TraceBuf& t, size_t words) {
    if (!t.dev) {
        (void)hipMalloc((void**)&t.dev, words * 8);
        t.words = words;
    }
    (void)hipMemset(t.dev, 0, t.words * 8);
    return t.dev;
}

// After launch number `dump_at` (counted from 0): print the stamps of waves 0 and 4 (per_wave slots each; a stamp = tag << 56 | time).
// spans > 0: behind the 8 wave areas sit `spans` (cycles, tiles) pairs, one per block -- printed as a summary.
inline void trace_end(TraceBuf& t, int dump_at, int per_wave, bool abs_time, unsigned spans) {
    (void)hipDeviceSynchronize();
    if (t.launches++ != dump_at) return;
    std::vector<unsigned long long> h(t.words);
    (void)hipMemcpy(h.data(), t.dev, t.words * 8, hipMemcpyDeviceToHost);
    if (spans) {
        unsigned long long mn = ~0ull, mx = 0, sum = 0;
        int nb = 0;
        unsigned long long xs[8] = {0}, xn[8] = {0};
        const size_t base = (size_t)8 * per_wave;
        for (unsigned b = 0; b < spans; ++b) {
            const unsigned long long v = h[base + 2 * b];
            if (!v) continue;
            mn = v < mn ? v : mn; mx = v > mx ? v : mx; sum += v; ++nb; xs[b & 7] += v; xn[b & 7]++;
        }
        fprintf(stderr, "BLOCKSPAN blocks %d min %llu avg %llu max %llu cycles; per XCD avg:", nb, mn, nb ? sum / nb : 0, mx);
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %llu", xn[x] ? xs[x] / xn[x] : 0);
        fprintf(stderr, "\n");
        for (unsigned b = 0; b < spans; b += 37) fprintf(stderr, "BLOCK %u span %llu tiles %llu\n", b, h[base + 2 * b], h[base + 2 * b + 1]);
    }
    for (int w = 0; w < 8; w += 4) {
        unsigned long long prev = 0;
        for (int k = 0; k < per_wave && h[(size_t)w * per_wave + k]; ++k) {
            const unsigned long long v = h[(size_t)w * per_wave + k], tm = v & 0x00ffffffffffffffull;
            if (abs_time) fprintf(stderr, "TR wave %d k %d tag %d t %llu d %lld\n", w, k, (int)(v >> 56), tm, prev ? (long long)(tm - prev) : 0ll);
            else fprintf(stderr, "TR wave %d k %d tag %d d %lld\n", w, k, (int)(v >> 56), prev ? (long long)(tm - prev) : 0ll);
            prev = tm;
        }
    }
}

}  // namespace seam_dev
