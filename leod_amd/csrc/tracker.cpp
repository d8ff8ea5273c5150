// Tracking post-filter of the pseudo-label loop -- host C++ (sequential, a few boxes per frame: not GPU work).
//
// Replaces modules/tracking/linear.py:10-292 (LinearBoxTracker, associate_tracking, LinearTracker),
// modules/tracking/utils.py:7-96 (greedy_matching, iou_batch_xywh, clamp_bbox) and EventSeqData._track
// (modules/pseudo_labeler.py:201-258), whose Python loops (`list.index`, per-box dict look-ups, deepcopy) are the
// sequential tail of every pseudo-labelling round once inference is fast.
//
// Arithmetic mirrors what the reference evaluates under NumPy >= 2: boxes and IoUs in float32 with one rounding per
// operation (this file is built with -ffp-contract=off), tracklet confidences in double, the IoU threshold compared
// as float32.  np.argsort of the confidences is a stable sort here (NumPy's order for more than 16 exactly tied values is
// implementation defined); everything else is bit-exact against the reference (tests/golden/g13_tracker.npz).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#define LEOD_HOST_API extern "C" __attribute__((visibility("default")))

namespace {

struct Box4 { float x, y, w, h; };              // centre format

struct Corners { float x1, y1, x2, y2; };

inline Corners corners(const Box4& b) {          // utils.py:65-72
    const float hw = b.w / 2.f, hh = b.h / 2.f;
    return {b.x - hw, b.y - hh, b.x + hw, b.y + hh};
}

struct Tracklet {
    Box4 box, last;                              // current / previous state (un-clamped), linear.py:33,71
    std::array<float, 5> pred;                   // clamped predicted state of the current frame (+ class)
    float cls;
    float vx = 0.f, vy = 0.f;
    bool clamp_t = false, clamp_d = false, clamp_l = false, clamp_r = false;
    bool is_gt, done = false;
    int age = 0, hits = 1;
    double conf;
    std::vector<int> boxes;                                          // global indices of the matched detections
    std::vector<std::pair<int, std::array<float, 5>>> missed, cache; // predicted boxes at frames without a match

    void predict(int img_h, int img_w) {         // linear.py:64-75 + get_state + utils.clamp_bbox(format_='xywh')
        age += 1;
        last = box;
        box.x += vx;
        box.y += vy;
        const Corners c = corners(box);
        const float W1 = (float)(img_w - 1), H1 = (float)(img_h - 1);
        const float x1 = std::fmin(std::fmax(c.x1, 0.f), W1), x2 = std::fmin(std::fmax(c.x2, 0.f), W1);
        const float y1 = std::fmin(std::fmax(c.y1, 0.f), H1), y2 = std::fmin(std::fmax(c.y2, 0.f), H1);
        clamp_t = y1 != c.y1; clamp_d = y2 != c.y2; clamp_l = x1 != c.x1; clamp_r = x2 != c.x2;
        pred = {(x1 + x2) / 2.f, (y1 + y2) / 2.f, x2 - x1, y2 - y1, cls};
    }

    void update(const float* det, int box_idx, bool det_is_gt, double q) {   // linear.py:77-98
        hits = age + 1;
        float nvx = det[0] - last.x, nvy = det[1] - last.y;
        if (clamp_t || clamp_d || clamp_l || clamp_r) {              // clamp-aware velocity, linear.py:100-122
            const Corners o = corners(last), n = corners(Box4{det[0], det[1], det[2], det[3]});
            if (clamp_t) nvy = n.y2 - o.y2;
            if (clamp_d) nvy = n.y1 - o.y1;
            if (clamp_l) nvx = n.x2 - o.x2;
            if (clamp_r) nvx = n.x1 - o.x1;
        }
        vx = nvx; vy = nvy;
        box = Box4{det[0], det[1], det[2], det[3]};
        boxes.push_back(box_idx);
        is_gt = is_gt || det_is_gt;
        const double w = q * (1.0 - std::pow(q, (double)age)) / (1.0 - q);
        conf = (w * conf + 1.0) / (w + 1.0);
        missed.insert(missed.end(), cache.begin(), cache.end());
        cache.clear();
    }
};

struct Tracker {
    int img_h, img_w;
    double min_conf, q;
    float iou_thr;
    std::vector<Tracklet> live, finished;        // finished in deletion order (= the reference's prev_trackers)
    int box_count = 0;

    void retire(size_t idx, bool done) {
        live[idx].done = done;
        finished.push_back(std::move(live[idx]));
        live.erase(live.begin() + idx);
    }

    void step(int frame, const float* dets, const unsigned char* gt, int n) {     // linear.py:213-284
        if (n == 0 && live.empty()) return;
        for (size_t t = live.size(); t-- > 0;)                        // zero-area tracklets are retired before matching
            if (live[t].box.w * live[t].box.h <= 0.f) retire(t, true);            // linear.py:235 `trk.area <= 0.`
        // NB the reference predicts the survivors first and deletes afterwards; both orders give the same state
        const int T = (int)live.size();
        std::vector<int> order(T);
        for (int t = 0; t < T; ++t) { live[t].predict(img_h, img_w); order[t] = t; }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return -live[a].conf < -live[b].conf; });
        std::vector<int> trk_of_det(n, -1), det_of_trk(T, -1);
        std::vector<std::pair<int, int>> matched;
        if (T > 0 && n > 0) {                                         // associate_tracking, linear.py:154-193
            std::vector<float> iou((size_t)T * n);
            bool positive = false, has_nan = false;
            for (int t = 0; t < T; ++t) {
                const auto& p = live[t].pred;
                const float tx1 = p[0] - p[2] / 2.f, ty1 = p[1] - p[3] / 2.f, tx2 = p[0] + p[2] / 2.f, ty2 = p[1] + p[3] / 2.f;
                const float ta = p[2] * p[3];
                for (int d = 0; d < n; ++d) {
                    const float* b = dets + 5 * d;
                    const float xx1 = std::fmax(tx1, b[0] - b[2] / 2.f), yy1 = std::fmax(ty1, b[1] - b[3] / 2.f);
                    const float xx2 = std::fmin(tx2, b[0] + b[2] / 2.f), yy2 = std::fmin(ty2, b[1] + b[3] / 2.f);
                    const float w = std::fmax(0.f, xx2 - xx1), h = std::fmax(0.f, yy2 - yy1);
                    const float wh = w * h;
                    float o = wh / (ta + b[2] * b[3] - wh);
                    if (b[4] != p[4]) o = 0.f;
                    iou[(size_t)t * n + d] = o;
                    positive = positive || o > 0.f;
                    has_nan = has_nan || std::isnan(o);
                }
            }
            if (positive && !has_nan) {                               // `iou_matrix.max() > 0` (NaN max -> False)
                const float ninf = -std::numeric_limits<float>::infinity();
                for (int t : order) {                                 // greedy_matching, utils.py:7-18
                    float best = ninf; int arg = 0;
                    for (int d = 0; d < n; ++d) {
                        const float v = iou[(size_t)t * n + d];
                        if (v > best) { best = v; arg = d; }
                    }
                    if (best < iou_thr) continue;
                    for (int tt = 0; tt < T; ++tt) iou[(size_t)tt * n + arg] = ninf;
                    matched.emplace_back(t, arg);
                    det_of_trk[t] = arg; trk_of_det[arg] = t;
                }
            }
        }
        bool any_gt = false;
        for (int d = 0; d < n; ++d) any_gt = any_gt || gt[d];
        for (const auto& m : matched) live[m.first].update(dets + 5 * m.second, box_count + m.second, gt[m.second] != 0, q);
        for (int t = 0; t < T; ++t)
            if (det_of_trk[t] < 0) {                                  // miss, linear.py:124-131
                live[t].conf *= q;
                if (!any_gt) live[t].cache.emplace_back(frame, live[t].pred);
            }
        for (int d = 0; d < n; ++d)
            if (trk_of_det[d] < 0) {
                Tracklet k;
                const float* b = dets + 5 * d;
                k.box = Box4{b[0], b[1], b[2], b[3]};
                k.last = k.box;
                k.pred = {b[0], b[1], b[2], b[3], b[4]};
                k.cls = b[4];
                k.is_gt = gt[d] != 0;
                k.conf = q;
                k.boxes.push_back(box_count + d);
                live.push_back(std::move(k));
            }
        for (size_t t = live.size(); t-- > 0;)
            if (live[t].conf < min_conf) retire(t, true);
        box_count += n;
    }

    void finish() {                                                   // tracker.py:35-40: unfinished tracklets are kept
        for (size_t t = live.size(); t-- > 0;) retire(t, false);
    }
};

}  // namespace

// Short-tracklet filter (+ optional in-painting) over one recording.
//   boxes [N,5] float32 (cx, cy, w, h, class) of the labelled frames, concatenated in frame order
//   is_gt [N], frame_idx [F] strictly increasing, counts [F] boxes per labelled frame
//   remove [N] out: 1 = the box sits on a finished, non-GT tracklet with fewer than min_track_len hits
//   inpaint != 0: the predicted boxes of the kept tracklets at the frames where they had no detection, in the reference's
//   iteration order: inp_frame [cap], inp_box [cap,5], *n_inp = number produced (also when cap is too small -> rc -3)
LEOD_HOST_API int leod_track_filter(const float* boxes, const unsigned char* is_gt, const int* frame_idx, const int* counts,
                                    int F, int img_h, int img_w, int min_track_len, double min_conf, double iou_threshold,
                                    double q, unsigned char* remove, int inpaint, int* inp_frame, float* inp_box,
                                    int inp_cap, int* n_inp) {
    if (F < 0 || img_h <= 0 || img_w <= 0 || (F > 0 && (!boxes || !is_gt || !frame_idx || !counts || !remove))) return -1;
    if (n_inp) *n_inp = 0;
    if (F == 0) return 0;
    for (int k = 1; k < F; ++k)
        if (frame_idx[k] <= frame_idx[k - 1]) return -1;
    if (frame_idx[0] < 0) return -1;
    Tracker trk{img_h, img_w, min_conf, q, (float)iou_threshold};
    int k = 0, off = 0;
    for (int f = 0; f <= frame_idx[F - 1]; ++f) {                     // pseudo_labeler.py:213-225: every frame, labelled or not
        if (k < F && frame_idx[k] == f) {
            trk.step(f, boxes + 5 * (size_t)off, is_gt + off, counts[k]);
            off += counts[k];
            ++k;
        } else {
            trk.step(f, nullptr, nullptr, 0);
        }
    }
    trk.finish();
    const int N = off;
    std::memset(remove, 0, (size_t)N);
    int produced = 0;
    for (const Tracklet& t : trk.finished) {
        const bool drop = t.done && !t.is_gt && t.hits < min_track_len;   // pseudo_labeler.py:231-236
        if (drop) {
            for (int b : t.boxes) remove[b] = 1;
            continue;
        }
        if (!inpaint) continue;
        for (const auto& m : t.missed) {                                  // pseudo_labeler.py:241-249
            if (produced < inp_cap && inp_frame && inp_box) {
                inp_frame[produced] = m.first;
                std::memcpy(inp_box + 5 * (size_t)produced, m.second.data(), 5 * sizeof(float));
            }
            ++produced;
        }
    }
    if (n_inp) *n_inp = produced;
    return (inpaint && produced > inp_cap) ? -3 : 0;
}
