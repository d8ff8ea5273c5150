// Detection evaluation of the validation / test loop -- host C++ (sequential per image, a few boxes each: not GPU work).
//
// Replaces what the reference delegates to pycocotools' COCOeval (or detectron2's C++ COCOeval_opt) from
// utils/evaluation/prophesee/metrics/coco_eval.py:121-139: per image and category greedy matching of score-ordered
// detections to ground truth at 10 IoU thresholds and 4 area ranges, then precision at 101 recall levels and the final
// recall for maxDets 1 / 10 / 100.  bbox mode, no crowd regions (the reference writes iscrowd False for every box,
// coco_eval.py:170), useCats = 1.  The Python side (leod_amd/utils/evaluation/prophesee/metrics/coco_eval.py) builds the
// image windows and averages the tables exactly as COCOeval.summarize does.
//
// Arithmetic: boxes are the float32 values of the Prophesee records widened to double (what np.array(.., dtype=double) of
// the COCO records holds), areas are float32 products (coco_eval.py:165 and COCO.loadRes), IoUs follow maskApi's bbIou.
// pycocotools itself is not in this image: parity for this file is unpinned beyond oracle/coco_eval.py (see its header).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#define LEOD_HOST_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kAreas = 4, kMaxDets = 3;
const double kAreaLo[kAreas] = {0.0, 0.0, 32.0 * 32.0, 96.0 * 96.0};
const double kAreaHi[kAreas] = {1e10, 32.0 * 32.0, 96.0 * 96.0, 1e10};
const int kMaxDet[kMaxDets] = {1, 10, 100};

struct ImageEval {                     // one (image, category, area range): COCOeval.evaluateImg
    std::vector<double> score;         // [D] descending, D <= 100
    std::vector<uint8_t> matched;      // [T*D]
    std::vector<uint8_t> ignored;      // [T*D]
    int n_gt_counted = 0;              // ground-truth boxes inside the area range
};

inline double box_iou(const float* d, const float* g) {            // bbIou, non-crowd
    const double dx = d[0], dy = d[1], dw = d[2], dh = d[3], gx = g[0], gy = g[1], gw = g[2], gh = g[3];
    const double w = std::min(dw + dx, gw + gx) - std::max(dx, gx);
    if (w <= 0) return 0.0;
    const double h = std::min(dh + dy, gh + gy) - std::max(dy, gy);
    if (h <= 0) return 0.0;
    const double inter = w * h;
    return inter / (dw * dh + gw * gh - inter);
}

}  // namespace

// gt_box [G,4] / dt_box [D,4] = (x, y, w, h) float32; *_off [n_img+1] = first row of every image window
// precision [T,R,K,4,3] and recall [T,K,4,3] are filled with -1 where COCOeval leaves them undefined.
LEOD_HOST_API int leod_coco_eval(const float* gt_box, const int* gt_cls, const int* gt_off, const float* dt_box,
                                 const int* dt_cls, const float* dt_score, const int* dt_off, int n_img, int n_cat,
                                 const double* iou_thrs, int T, const double* rec_thrs, int R, double* precision,
                                 double* recall) {
    if (n_img < 0 || n_cat <= 0 || T <= 0 || R <= 0 || !iou_thrs || !rec_thrs || !precision || !recall || !gt_off || !dt_off)
        return -1;
    const int G = gt_off[n_img], D = dt_off[n_img];
    if ((G > 0 && (!gt_box || !gt_cls)) || (D > 0 && (!dt_box || !dt_cls || !dt_score))) return -1;
    const size_t K = (size_t)n_cat;
    std::fill(precision, precision + (size_t)T * R * K * kAreas * kMaxDets, -1.0);
    std::fill(recall, recall + (size_t)T * K * kAreas * kMaxDets, -1.0);
    const int cap = kMaxDet[kMaxDets - 1];

    std::vector<int> gi, di, gorder;
    std::vector<double> iou;
    std::vector<uint8_t> g_ig, g_taken;
    for (int k = 0; k < n_cat; ++k) {
        std::vector<ImageEval> evals[kAreas];
        for (int img = 0; img < n_img; ++img) {
            gi.clear(); di.clear();
            for (int j = gt_off[img]; j < gt_off[img + 1]; ++j) if (gt_cls[j] == k) gi.push_back(j);
            for (int j = dt_off[img]; j < dt_off[img + 1]; ++j) if (dt_cls[j] == k) di.push_back(j);
            if (gi.empty() && di.empty()) continue;
            std::stable_sort(di.begin(), di.end(), [&](int a, int b) { return dt_score[a] > dt_score[b]; });
            if ((int)di.size() > cap) di.resize(cap);
            const int nd = (int)di.size(), ng = (int)gi.size();
            iou.assign((size_t)nd * ng, 0.0);
            for (int d = 0; d < nd; ++d)
                for (int g = 0; g < ng; ++g) iou[(size_t)d * ng + g] = box_iou(dt_box + 4 * (size_t)di[d], gt_box + 4 * (size_t)gi[g]);
            for (int a = 0; a < kAreas; ++a) {
                ImageEval ev;
                g_ig.assign(ng, 0);
                for (int g = 0; g < ng; ++g) {
                    const float area = gt_box[4 * (size_t)gi[g] + 2] * gt_box[4 * (size_t)gi[g] + 3];
                    g_ig[g] = ((double)area < kAreaLo[a] || (double)area > kAreaHi[a]) ? 1 : 0;
                }
                gorder.resize(ng);
                std::iota(gorder.begin(), gorder.end(), 0);
                std::stable_sort(gorder.begin(), gorder.end(), [&](int x, int y) { return g_ig[x] < g_ig[y]; });
                ev.score.resize(nd);
                for (int d = 0; d < nd; ++d) ev.score[d] = (double)dt_score[di[d]];
                ev.matched.assign((size_t)T * nd, 0);
                ev.ignored.assign((size_t)T * nd, 0);
                for (int g = 0; g < ng; ++g) ev.n_gt_counted += g_ig[g] == 0;
                for (int t = 0; t < T; ++t) {
                    g_taken.assign(ng, 0);
                    for (int d = 0; d < nd; ++d) {
                        double best = std::min(iou_thrs[t], 1 - 1e-10);
                        int m = -1;
                        for (int s = 0; s < ng; ++s) {                 // s walks the ignore-sorted ground truth
                            const int g = gorder[s];
                            if (g_taken[g]) continue;
                            if (m > -1 && g_ig[m] == 0 && g_ig[g] == 1) break;
                            const double v = iou[(size_t)d * ng + g];
                            if (v < best) continue;
                            best = v;
                            m = g;
                        }
                        if (m == -1) continue;
                        g_taken[m] = 1;
                        ev.matched[(size_t)t * nd + d] = 1;
                        ev.ignored[(size_t)t * nd + d] = g_ig[m];
                    }
                }
                for (int d = 0; d < nd; ++d) {                         // unmatched detections outside the area range
                    const float area = dt_box[4 * (size_t)di[d] + 2] * dt_box[4 * (size_t)di[d] + 3];
                    if ((double)area < kAreaLo[a] || (double)area > kAreaHi[a])
                        for (int t = 0; t < T; ++t)
                            if (!ev.matched[(size_t)t * nd + d]) ev.ignored[(size_t)t * nd + d] = 1;
                }
                evals[a].push_back(std::move(ev));
            }
        }
        // COCOeval.accumulate
        std::vector<double> sc, pr, rc;
        std::vector<int> order, src_img, src_det;
        for (int a = 0; a < kAreas; ++a) {
            const auto& E = evals[a];
            if (E.empty()) continue;
            long npig = 0;
            for (const auto& e : E) npig += e.n_gt_counted;
            if (npig == 0) continue;
            for (int m = 0; m < kMaxDets; ++m) {
                sc.clear(); src_img.clear(); src_det.clear();
                for (size_t i = 0; i < E.size(); ++i) {
                    const int n = std::min((int)E[i].score.size(), kMaxDet[m]);
                    for (int d = 0; d < n; ++d) { sc.push_back(E[i].score[d]); src_img.push_back((int)i); src_det.push_back(d); }
                }
                const int nd = (int)sc.size();
                order.resize(nd);
                std::iota(order.begin(), order.end(), 0);
                std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sc[x] > sc[y]; });
                pr.resize(nd); rc.resize(nd);
                for (int t = 0; t < T; ++t) {
                    double tp = 0, fp = 0;
                    for (int j = 0; j < nd; ++j) {
                        const auto& e = E[src_img[order[j]]];
                        const size_t at = (size_t)t * e.score.size() + src_det[order[j]];
                        const bool ig = e.ignored[at], hit = e.matched[at];
                        tp += (hit && !ig) ? 1.0 : 0.0;
                        fp += (!hit && !ig) ? 1.0 : 0.0;
                        rc[j] = tp / (double)npig;
                        pr[j] = tp / (fp + tp + 2.220446049250313e-16);
                    }
                    recall[(((size_t)t * K + k) * kAreas + a) * kMaxDets + m] = nd ? rc[nd - 1] : 0.0;
                    for (int j = nd - 1; j > 0; --j)
                        if (pr[j] > pr[j - 1]) pr[j - 1] = pr[j];
                    for (int r = 0; r < R; ++r) {
                        const int pi = (int)(std::lower_bound(rc.begin(), rc.begin() + nd, rec_thrs[r]) - rc.begin());
                        precision[((((size_t)t * R + r) * K + k) * kAreas + a) * kMaxDets + m] = pi < nd ? pr[pi] : 0.0;
                    }
                }
            }
        }
    }
    return 0;
}
