// seam_tracklets.hip -- HOST helper of the evaluator (no device code): the greedy tracklet linking of
// evaluate_movingfashion.py:166-202 for all products of a pass in one call.
//
// The linking is a chain of tiny data-dependent decisions over a product's <= a few dozen detections (seed with the most
// confident free detection; repeatedly take, among the free detections of frames that have no member yet, the one most similar to
// ANY current member; link it if that similarity exceeds the threshold, else close the tracklet) -- nothing for a GPU, but in
// Python / NumPy it was what the batched evaluator spent its time on (~0.2 ms per product of interpreter and array-call overhead
// against ~0.02 ms of device work).  Same decisions as the reference's loops, element for element: candidates in ascending
// detection order, rows (current members) in ascending detection order, the FIRST maximum in row-major order wins (np.argmax).
#include <stdint.h>
#include <vector>
#include <algorithm>

extern "C" {

// blocks: the products' n_s x n_s self-similarity blocks, concatenated (offset of product s = sum of n_t^2, t < s);
// seg [n_seg + 1]: detection offsets; imgs / scores [seg[n_seg]]: frame index and confidence of every detection;
// members_out [seg[n_seg]]: per product (at seg[s]) its detections' LOCAL indices, tracklet after tracklet in creation order, members
// in link order; track_len_out [seg[n_seg]]: per product (at seg[s]) the lengths of its tracklets; n_tracks_out [n_seg].  Returns 0.
int seam_host_build_tracklets(const float* blocks, const int64_t* seg, const int64_t* imgs, const double* scores, int n_seg,
                              double threshold, int32_t* members_out, int32_t* track_len_out, int32_t* n_tracks_out) {
    std::vector<char> free_, open_, present;
    std::vector<int> members, sorted;
    size_t boff = 0;
    for (int s = 0; s < n_seg; ++s) {
        const int64_t o = seg[s];
        const int n = (int)(seg[s + 1] - o);
        const float* sim = blocks + boff;
        boff += (size_t)n * n;
        const int64_t* im = imgs + o;
        const double* sc = scores + o;
        int nfr = 0;
        for (int i = 0; i < n; ++i) nfr = std::max(nfr, (int)im[i] + 1);
        free_.assign(n, 1);
        present.assign(nfr, 0);
        for (int i = 0; i < n; ++i) present[im[i]] = 1;
        int left = n, nt = 0, mo = 0;
        while (left > 0) {
            int start = -1;
            for (int i = 0; i < n; ++i)
                if (free_[i] && (start < 0 || sc[i] > sc[start])) start = i;            // first maximum among the free detections
            members.assign(1, start);
            sorted.assign(1, start);
            open_ = present;
            open_[im[start]] = 0;
            for (;;) {
                float best = 0.f;
                int best_c = -1;
                for (int r : sorted)                                                     // rows in detection order
                    for (int j = 0; j < n; ++j)
                        if (free_[j] && open_[im[j]]) {
                            const float v = sim[(size_t)r * n + j];
                            if (best_c < 0 || v > best) { best = v; best_c = j; }      // strict: the first maximum in row-major order
                        }
                // (the comparison in fp32, as `sub[r, c] > threshold` is with a float32 matrix and a Python float under NumPy 2 promotion)
                if (best_c < 0 || !(best > (float)threshold)) break;
                members.push_back(best_c);
                sorted.insert(std::upper_bound(sorted.begin(), sorted.end(), best_c), best_c);
                open_[im[best_c]] = 0;
            }
            for (int m : members) { free_[m] = 0; members_out[o + mo++] = m; }
            left -= (int)members.size();
            track_len_out[o + nt++] = (int)members.size();
        }
        n_tracks_out[s] = nt;
    }
    return 0;
}

}  // extern "C"
