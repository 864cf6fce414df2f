// eval_host.hip -- host half of the KITTI evaluation: the greedy ground-truth <-> detection matching
// that the reference runs as numba CPU code, `compute_statistics_jit` (mmdet/core/evaluation/kitti_eval.py:165-283) and
// its loop over images x score thresholds `fused_compute_statistics` (:296-343).  It is sequential per image (every
// match removes a detection from the pool) and tiny next to the O(N*K) rotated-IoU matrices, which stay on the GPU
// (eval.hip); so this file is plain C++ behind the same C ABI, and touches no device memory.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/sassd.h"

namespace {
constexpr double kNone = -10000000.0;          // "no detection picked yet" (kitti_eval.py:191)

struct Image {
    const double *ov;                          // [n_dt rows, ld] -- overlap of detection j with ground truth i at ov[j*ld+i]
    int64_t ld;
    int64_t n_gt, n_dt, n_dc;
    const double *dt;                          // [n_dt,6]  bbox(4), alpha, score
    const double *gt;                          // [n_gt,5]  bbox(4), alpha
    const double *dc;                          // [n_dc,4]  DontCare image boxes
    const int64_t *ign_gt, *ign_dt;            // 0 = counts, 1 = neutral, -1 = other class
};

struct Tally { int64_t tp = 0, fp = 0, fn = 0; double similarity = 0.0; };

// share of detection box d covered by box q (image_box_overlap with criterion 0, kitti_eval.py:96-122)
inline double covered_fraction(const double *d, const double *q)
{
    const double iw = std::fmin(d[2], q[2]) - std::fmax(d[0], q[0]);
    if (!(iw > 0)) return 0.0;
    const double ih = std::fmin(d[3], q[3]) - std::fmax(d[1], q[1]);
    if (!(ih > 0)) return 0.0;
    return iw * ih / ((d[2] - d[0]) * (d[3] - d[1]));
}

// One image at one score threshold.  count_fp = false is the first pass of the evaluation (every detection is a
// candidate, the best-SCORING one over min_overlap wins, only the scores of the true positives are wanted);
// count_fp = true is the precision/recall pass (detections under `thresh` are out, the best-OVERLAPPING counted
// detection wins, a neutral detection is taken only when nothing else was found).
Tally match_image(const Image &im, int metric, double min_overlap, double thresh, bool count_fp, bool want_aos,
                  std::vector<char> &taken, std::vector<char> &below, std::vector<double> &delta, double *tp_scores,
                  int64_t *n_tp_scores)
{
    Tally t;
    taken.assign((size_t)im.n_dt, 0);
    below.assign((size_t)im.n_dt, 0);
    delta.clear();
    if (count_fp)
        for (int64_t j = 0; j < im.n_dt; ++j) below[j] = im.dt[j * 6 + 5] < thresh;

    for (int64_t i = 0; i < im.n_gt; ++i) {
        if (im.ign_gt[i] == -1) continue;
        int64_t pick = -1;
        double picked = kNone, best_ov = 0.0;
        bool pick_is_neutral = false;
        for (int64_t j = 0; j < im.n_dt; ++j) {
            if (im.ign_dt[j] == -1 || taken[j] || below[j]) continue;
            const double ov = im.ov[j * im.ld + i];
            if (!(ov > min_overlap)) continue;
            if (!count_fp) {
                const double s = im.dt[j * 6 + 5];
                if (s > picked) { pick = j; picked = s; }
            } else if ((ov > best_ov || pick_is_neutral) && im.ign_dt[j] == 0) {
                best_ov = ov; pick = j; picked = 1.0; pick_is_neutral = false;
            } else if (picked == kNone && im.ign_dt[j] == 1) {
                pick = j; picked = 1.0; pick_is_neutral = true;
            }
        }
        const bool found = picked != kNone;
        if (!found) {
            if (im.ign_gt[i] == 0) ++t.fn;
        } else if (im.ign_gt[i] == 1 || im.ign_dt[pick] == 1) {
            taken[pick] = 1;                                       // neutral pairing: neither a hit nor a miss
        } else {
            ++t.tp;
            if (tp_scores) tp_scores[(*n_tp_scores)++] = im.dt[pick * 6 + 5];
            if (want_aos) delta.push_back(im.gt[i * 5 + 4] - im.dt[pick * 6 + 4]);
            taken[pick] = 1;
        }
    }
    if (!count_fp) return t;

    for (int64_t j = 0; j < im.n_dt; ++j)
        if (!(taken[j] || im.ign_dt[j] != 0 || below[j])) ++t.fp;
    if (metric == 0) {                                             // detections sitting in DontCare regions are forgiven
        int64_t forgiven = 0;
        for (int64_t c = 0; c < im.n_dc; ++c)
            for (int64_t j = 0; j < im.n_dt; ++j) {
                if (taken[j] || im.ign_dt[j] != 0 || below[j]) continue;
                if (covered_fraction(im.dt + j * 6, im.dc + c * 4) > min_overlap) { taken[j] = 1; ++forgiven; }
            }
        t.fp -= forgiven;
    }
    if (want_aos) {
        if (t.tp > 0 || t.fp > 0) {
            double s = 0.0;
            for (double d : delta) s += (1.0 + std::cos(d)) / 2.0;
            t.similarity = s;
        } else {
            t.similarity = -1.0;                                   // "nothing to average" marker of the reference
        }
    }
    return t;
}
}  // namespace

extern "C" int sassd_kitti_eval_statistics(const double *overlaps, int64_t ld, int n_img, const int64_t *gt_nums,
                                           const int64_t *dt_nums, const int64_t *dc_nums, const double *gt_datas,
                                           const double *dt_datas, const double *dontcares, const int64_t *ignored_gts,
                                           const int64_t *ignored_dets, int metric, double min_overlap,
                                           const double *thresholds, int n_thr, int compute_aos, double *pr,
                                           double *tp_scores, int64_t *n_tp_scores)
{
    if (n_img < 0 || n_thr < 0 || ld < 0 || metric < 0 || metric > 2) return SASSD_EINVAL;
    if (n_img > 0 && (!gt_nums || !dt_nums || !dc_nums)) return SASSD_EINVAL;
    if (n_thr > 0 && (!thresholds || !pr)) return SASSD_EINVAL;
    if (n_thr == 0 && (!tp_scores || !n_tp_scores)) return SASSD_EINVAL;
    int64_t sum_gt = 0;
    for (int i = 0; i < n_img; ++i) {
        if (gt_nums[i] < 0 || dt_nums[i] < 0 || dc_nums[i] < 0) return SASSD_EINVAL;
        sum_gt += gt_nums[i];
    }
    if (sum_gt > ld) return SASSD_EINVAL;
    if (n_thr == 0) *n_tp_scores = 0;

    std::vector<char> taken, below;
    std::vector<double> delta;
    int64_t g0 = 0, d0 = 0, c0 = 0;
    for (int i = 0; i < n_img; ++i) {
        Image im;
        im.ld = ld;
        im.n_gt = gt_nums[i]; im.n_dt = dt_nums[i]; im.n_dc = dc_nums[i];
        im.ov = overlaps + d0 * ld + g0;                           // the image's diagonal block of the part matrix
        im.gt = gt_datas + g0 * 5; im.dt = dt_datas + d0 * 6; im.dc = dontcares + c0 * 4;
        im.ign_gt = ignored_gts + g0; im.ign_dt = ignored_dets + d0;
        if (n_thr == 0) {
            const Tally r = match_image(im, metric, min_overlap, 0.0, false, false, taken, below, delta, tp_scores,
                                        n_tp_scores);
            if (pr) { pr[0] += (double)r.tp; pr[2] += (double)r.fn; }
        } else {
            for (int t = 0; t < n_thr; ++t) {
                const Tally r = match_image(im, metric, min_overlap, thresholds[t], true, compute_aos != 0, taken,
                                            below, delta, nullptr, nullptr);
                pr[t * 4 + 0] += (double)r.tp;
                pr[t * 4 + 1] += (double)r.fp;
                pr[t * 4 + 2] += (double)r.fn;
                if (r.similarity != -1.0) pr[t * 4 + 3] += r.similarity;
            }
        }
        g0 += im.n_gt; d0 += im.n_dt; c0 += im.n_dc;
    }
    return SASSD_OK;
}
