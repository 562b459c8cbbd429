// seam_topk.h -- workgroup-level exact top-k under the ranking order of the evaluator (device code shared by seam_heads.hip
// and seam_pairmf.hip).
//
// Order (ref evaluate_movingfashion.py:97-99,265-269: softmax(x5)[...,1] descending): d = x1 - x0 descending (monotone in the
// score, no softmax saturation ties), index ascending on equal d (the reference's reversed unstable argsort leaves ties
// unspecified), NaN last.
//
// Exact top-k of one row of n (x0, x1) pairs, O(n) instead of k arg-max rounds: an MSB-first 8-bit radix select over
// order-preserving keys finds the k-th largest key T (4 histogram passes in LDS), everything above T is collected with one pass,
// ties at T are taken lowest-index-first, and the k winners are ordered by rank counting.
// Item j of the row is (x0, x1, g) = load(j); g is the value reported as its index (g < 0: the item does not exist).
// All functions are called by a whole 256-thread workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace seam_topk {

__device__ __forceinline__ unsigned tk_key_d(float d) {
    if (d != d) d = -INFINITY;                   // NaN ranks last
    d += 0.f;                                    // -0 -> +0 (equal scores must tie)
    const unsigned u = __float_as_uint(d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned tk_key(float x0, float x1) { return tk_key_d(x1 - x0); }

// softmax(x)[1] of a logit pair (ref evaluate_movingfashion.py:97-98)
__device__ __forceinline__ float tk_score(float x0, float x1) {
    const float mx = fmaxf(x0, x1);
    const float e0 = expf(x0 - mx), e1 = expf(x1 - mx);
    return e1 / (e0 + e1);
}

struct TopkShared {
    unsigned hist[256];
    unsigned key[256];
    int item[256];
    int gidx[256];
    unsigned prefix, krem, cnt;
    int red[4];
};

// histogram increment with wave aggregation: when every participating lane of the wave hits the SAME bin (the leading radix
// digits of a row of similar scores), one lane adds the population count instead of 64 serialised LDS atomics
__device__ __forceinline__ void tk_hist_add(unsigned* hist, bool act, unsigned bin) {
    const unsigned long long m = __ballot(act);
    if (m == 0ull) return;
    const int first = __ffsll((long long)m) - 1;
    const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)bin, first);      // wave-uniform lane index: no LDS round trip
    if (__ballot(act && bin != b0) == 0ull) {
        if ((int)(threadIdx.x & 63) == first) atomicAdd(&hist[b0], (unsigned)__popcll(m));
    } else if (act) {
        atomicAdd(&hist[bin], 1u);
    }
}

// k-th largest key T of the row (1 <= k <= number of existing items); krem (>= 1) = how many of the items equal to T
// belong to the top k.
template <typename Load>
__device__ void block_kth(Load load, int n, int k, TopkShared& sh, unsigned& T, unsigned& krem_out) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    unsigned prefix = 0, mask = 0, krem = (unsigned)k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        sh.hist[tid] = 0;
        __syncthreads();
        for (int j0 = 0; j0 < n; j0 += 256) {          // whole waves stay in the loop: tk_hist_add is a wave operation
            const int j = j0 + tid;
            bool act = false;
            unsigned bin = 0;
            if (j < n) {
                float x0, x1; int g;
                load(j, x0, x1, g);
                const unsigned key = g < 0 ? 0u : tk_key(x0, x1);
                act = g >= 0 && (key & mask) == prefix;
                bin = (key >> shift) & 255u;
            }
            tk_hist_add(sh.hist, act, bin);
        }
        __syncthreads();
        // digit selection, all 256 threads: thread t owns bin 255 - t, an inclusive scan over t counts the keys in bins >= its own
        const unsigned c = sh.hist[255 - tid];
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) sh.red[wid] = (int)incl;
        __syncthreads();
        unsigned base = 0;
        for (int w2 = 0; w2 < wid; ++w2) base += (unsigned)sh.red[w2];
        incl += base;
        const unsigned excl = incl - c;
        if (excl < krem && krem <= incl) {              // exactly one bin straddles the k-th place
            sh.prefix = prefix | ((unsigned)(255 - tid) << shift);
            sh.krem = krem - excl;
        } else if (tid == 255 && incl < krem) {         // fewer than k existing items (callers avoid it): lowest bin, as a serial scan would
            sh.prefix = prefix;
            sh.krem = krem - incl;
        }
        __syncthreads();
        prefix = sh.prefix;
        krem = sh.krem;
        mask |= 0xFFu << shift;
    }
    T = prefix;
    krem_out = krem;
}

// ---- ONE wave, no barriers (LDS operations of a wave execute in order): the k-th largest of the keys its lanes hold in registers.
// key[i] of lane l is item i * 64 + l of nvalid items; hist = 256 words of LDS owned by this wave.  Returns T; krem as above.
template <int PER>
__device__ __forceinline__ unsigned wave_kth(const unsigned (&key)[PER], int nvalid, int k, unsigned* hist, unsigned& krem_out) {
    const int lane = threadIdx.x & 63;
    unsigned prefix = 0, mask = 0, krem = (unsigned)k;
    for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) hist[lane + 64 * i] = 0u;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const bool act = i * 64 + lane < nvalid && (key[i] & mask) == prefix;
            tk_hist_add(hist, act, (key[i] >> shift) & 255u);
        }
        // lane l owns bins 255 - 4 l .. 252 - 4 l (descending); inclusive scan of the lane sums over the wave
        unsigned c[4], mine = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { c[i] = hist[255 - 4 * lane - i]; mine += c[i]; }
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += up;
        }
        unsigned run = incl - mine;                  // keys in bins above this lane's
        int bsel = -1;
        unsigned kr = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (bsel < 0 && run < krem && krem <= run + c[i]) { bsel = 255 - 4 * lane - i; kr = krem - run; }
            run += c[i];
        }
        const unsigned long long who = __ballot(bsel >= 0);
        const int src = who ? __ffsll((long long)who) - 1 : 63;
        const int bs = __builtin_amdgcn_readlane(bsel, src);
        const unsigned ks = (unsigned)__builtin_amdgcn_readlane((int)kr, src);
        if (who) { prefix |= (unsigned)bs << shift; krem = ks; }
        else { krem -= __builtin_amdgcn_readlane((int)incl, 63); }   // fewer than k items: lowest bin, as the block version
        mask |= 0xFFu << shift;
    }
    krem_out = krem;
    return prefix;
}

// The k winners, unordered, into sh.key / sh.item / sh.gidx [0, k): everything above T, then the krem lowest-index items at T.
template <typename Load>
__device__ void block_collect(Load load, int n, int k, unsigned T, unsigned krem, TopkShared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nabove = k - (int)krem;
    if (tid == 0) sh.cnt = 0;
    __syncthreads();
    for (int j = tid; j < n; j += 256) {
        float x0, x1; int g;
        load(j, x0, x1, g);
        if (g < 0) continue;
        const unsigned key = tk_key(x0, x1);
        if (key > T) {
            const unsigned pos = atomicAdd(&sh.cnt, 1u);
            sh.key[pos] = key; sh.item[pos] = j; sh.gidx[pos] = g;
        }
    }
    __syncthreads();
    int last = -1;
    for (unsigned r = 0; r < krem; ++r) {        // ties at T: lowest reported index first (usually one round)
        int best = 0x7fffffff, bestj = -1;
        for (int j = tid; j < n; j += 256) {
            float x0, x1; int g;
            load(j, x0, x1, g);
            if (g > last && g < best && tk_key(x0, x1) == T) { best = g; bestj = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ob = __shfl_xor(best, o, 64), oj = __shfl_xor(bestj, o, 64);
            if (ob < best) { best = ob; bestj = oj; }
        }
        if (lane == 0) { sh.red[wid] = best; sh.hist[wid] = (unsigned)bestj; }
        __syncthreads();
        if (tid == 0) {
            int bb = sh.red[0], bj = (int)sh.hist[0];
            for (int w2 = 1; w2 < 4; ++w2)
                if (sh.red[w2] < bb) { bb = sh.red[w2]; bj = (int)sh.hist[w2]; }
            sh.key[nabove + r] = T; sh.item[nabove + r] = bj; sh.gidx[nabove + r] = bb;
            sh.red[0] = bb;
        }
        __syncthreads();
        last = sh.red[0];
        __syncthreads();
    }
}

// rank of winner `t` (0 <= t < k) among the k collected winners under (key desc, index asc)
__device__ __forceinline__ int block_winner_rank(const TopkShared& sh, int t, int k) {
    const unsigned mk = sh.key[t];
    const int mg = sh.gidx[t];
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < k; ++j) rank += (sh.key[j] > mk || (sh.key[j] == mk && sh.gidx[j] < mg)) ? 1 : 0;
    return rank;
}

template <typename Load>
__device__ void block_topk(Load load, int n, int k, int64_t* __restrict__ idx_out, float* __restrict__ score_out,
                           TopkShared& sh) {
    const int tid = threadIdx.x;
    unsigned T, krem;
    block_kth(load, n, k, sh, T, krem);
    block_collect(load, n, k, T, krem, sh);
    if (tid < k) {                               // order the k winners by counting (k <= 256)
        const int mg = sh.gidx[tid], mj = sh.item[tid];
        const int rank = block_winner_rank(sh, tid, k);
        float sc = 0.f;
        if (mj >= 0) {
            float x0, x1; int g;
            load(mj, x0, x1, g);
            sc = tk_score(x0, x1);
        }
        idx_out[rank] = mj >= 0 ? (int64_t)mg : (int64_t)-1;
        score_out[rank] = sc;
    }
}

}  // namespace seam_topk
