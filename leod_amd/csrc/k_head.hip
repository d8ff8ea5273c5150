// YOLOX head tail for the LEOD path: prediction 1x1 convs + grid decode, SimOTA label assignment,
// IoU/BCE(/focal) losses with their gradients, batched NMS (torchvision semantics) and the
// pseudo-label box filters.  Integer results (assignment masks/indices, NMS keep order) are meant to be
// bit-identical to the reference, so this file is compiled with fp contraction OFF: every fp32
// operation is rounded separately, in the reference's evaluation order.
//
// Reference: models/detection/yolox/models/yolo_head.py:208-332 (preds, decode), :403-774 and :776-1148
// (losses, SimOTA, ignore variant), models/detection/yolox/models/losses.py:18-85,
// models/detection/yolox/utils/boxes.py:32-113 (postprocess, IoU), modules/utils/ssod.py:40-188.
#include "common.hpp"
#pragma clang fp contract(off)

#define MAXLVL 3
struct Levels { int n; int h[MAXLVL], w[MAXLVL], stride[MAXLVL], a0[MAXLVL]; int A; };

__device__ __forceinline__ void anchor_geom(const Levels& L, int a, float& gx, float& gy, float& gs) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < MAXLVL; ++k) if (k < L.n && a >= L.a0[k]) l = k;
    const int idx = a - L.a0[l];
    const int y = idx / L.w[l];
    gy = (float)y; gx = (float)(idx - y * L.w[l]); gs = (float)L.stride[l];
}

// ---------------------------------------------------------------------------------------------------
// prediction convs + decode for one FPN level.  feat maps are NHWC [B, h*w, Hd].
//   raw = (reg(4), obj(1), cls(nc)) ; xy = (raw_xy + grid)*stride ; wh = exp(raw_wh)*stride
//   out_train: decoded boxes + logits ; out_infer: decoded boxes + sigmoid probabilities
// ---------------------------------------------------------------------------------------------------
// Feature rows staged through LDS (Hd <= 128): a workgroup takes 64 positions, copies their reg / cls rows with coalesced 16-byte loads
// (row stride Hd + 4 floats: the rows a wave reads next fall on different banks) and then computes its 64 x (5 + nc) outputs, one per thread and
// round, from LDS.  The per-(position, channel) kernel below reads the rows straight from global memory: neighbouring lanes are 384 bytes apart
// and every row is fetched by 5 + 2 lanes -- 424 us for the 860 k positions of level 0 in the pseudo-label pass (1.5 TB/s); a per-position variant
// without LDS was slower still (689 us: profiles/r04_a_graph_ab.txt).  Same fmaf chain per output (k ascending): the values are unchanged.
template <int HDMAX>
__global__ __launch_bounds__(256) void head_pred_fwd_lds_kernel(const float* __restrict__ cls_feat, const float* __restrict__ reg_feat,
                                                                const float* __restrict__ cls_w, const float* __restrict__ cls_b,
                                                                const float* __restrict__ reg_w, const float* __restrict__ reg_b,
                                                                const float* __restrict__ obj_w, const float* __restrict__ obj_b,
                                                                float* __restrict__ out_train, float* __restrict__ out_infer,
                                                                int B, int hw, int wl, int Hd, int nc, int stride, int a0, int A) {
    constexpr int TP = 64, LDR = HDMAX + 4, MAXCH = 5 + 16;
    __shared__ __attribute__((aligned(16))) float sR[TP * LDR];
    __shared__ __attribute__((aligned(16))) float sC[TP * LDR];
    __shared__ __attribute__((aligned(16))) float sW[MAXCH * HDMAX];     // the 5 + nc weight rows (a global load per k step was a latency chain)
    __shared__ float sBias[MAXCH];
    const int nch = 5 + nc, tid = threadIdx.x;
    const long total = (long)B * hw;
    const int k4n = Hd / 4;
    for (int e = tid; e < nch * k4n; e += 256) {
        const int ch = e / k4n, k4 = e - ch * k4n;
        const float* w = ch < 4 ? reg_w + (long)ch * Hd : ch == 4 ? obj_w : cls_w + (long)(ch - 5) * Hd;
        *reinterpret_cast<f4*>(&sW[ch * HDMAX + 4 * k4]) = ld4(w + 4 * k4);
    }
    if (tid < nch) sBias[tid] = tid < 4 ? reg_b[tid] : tid == 4 ? obj_b[0] : cls_b[tid - 5];
    for (long p0 = (long)blockIdx.x * TP; p0 < total; p0 += (long)gridDim.x * TP) {
        const int np = (int)min((long)TP, total - p0);
        __syncthreads();                                  // the previous tile's readers are done
        for (int e = tid; e < np * k4n; e += 256) {
            const int r = e / k4n, k4 = e - r * k4n;
            *reinterpret_cast<f4*>(&sR[r * LDR + 4 * k4]) = ld4(reg_feat + (p0 + r) * Hd + 4 * k4);
            *reinterpret_cast<f4*>(&sC[r * LDR + 4 * k4]) = ld4(cls_feat + (p0 + r) * Hd + 4 * k4);
        }
        __syncthreads();
        for (int o = tid; o < np * nch; o += 256) {
            const int r = o / nch, ch = o - r * nch;
            const long pos = p0 + r;
            const int p = (int)(pos % hw), b = (int)(pos / hw);
            const float* f = (ch < 5 ? sR : sC) + r * LDR;
            const float* w = sW + ch * HDMAX;
            float acc = 0.f;
#pragma unroll 4
            for (int k = 0; k < Hd; k += 4) {
                const f4 a = *reinterpret_cast<const f4*>(f + k), ww = *reinterpret_cast<const f4*>(w + k);
                acc = fmaf(a.x, ww.x, acc); acc = fmaf(a.y, ww.y, acc); acc = fmaf(a.z, ww.z, acc); acc = fmaf(a.w, ww.w, acc);
            }
            acc += sBias[ch];
            float vt = acc, vi = acc;
            if (ch < 2) { const int gy = p / wl, gx = p - gy * wl; vt = vi = (acc + (float)(ch == 0 ? gx : gy)) * (float)stride; }
            else if (ch < 4) { vt = vi = expf(acc) * (float)stride; }
            else { vi = sigmoidf_(acc); }
            const long oo = ((long)b * A + a0 + p) * nch + ch;
            if (out_train) out_train[oo] = vt;
            if (out_infer) out_infer[oo] = vi;
        }
    }
}
__global__ __launch_bounds__(256) void head_pred_fwd_kernel(const float* __restrict__ cls_feat, const float* __restrict__ reg_feat,
                                                            const float* __restrict__ cls_w, const float* __restrict__ cls_b,
                                                            const float* __restrict__ reg_w, const float* __restrict__ reg_b,
                                                            const float* __restrict__ obj_w, const float* __restrict__ obj_b,
                                                            float* __restrict__ out_train, float* __restrict__ out_infer,
                                                            int B, int hw, int wl, int Hd, int nc, int stride, int a0, int A) {
    const int nch = 5 + nc;
    const long total = (long)B * hw * nch;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % nch);
        const long pos = idx / nch;                 // b*hw + p
        const int p = (int)(pos % hw);
        const int b = (int)(pos / hw);
        const float* f; const float* w; float bias;
        if (ch < 4) { f = reg_feat; w = reg_w + (long)ch * Hd; bias = reg_b[ch]; }
        else if (ch == 4) { f = reg_feat; w = obj_w; bias = obj_b[0]; }
        else { f = cls_feat; w = cls_w + (long)(ch - 5) * Hd; bias = cls_b[ch - 5]; }
        f += pos * Hd;
        float acc = 0.f;
        for (int k = 0; k < Hd; k += 4) {
            const f4 a = ld4(f + k), ww = ld4(w + k);
            acc = fmaf(a.x, ww.x, acc); acc = fmaf(a.y, ww.y, acc); acc = fmaf(a.z, ww.z, acc); acc = fmaf(a.w, ww.w, acc);
        }
        acc += bias;
        float vt = acc, vi = acc;
        if (ch < 2) { const int gy = p / wl, gx = p - gy * wl; vt = vi = (acc + (float)(ch == 0 ? gx : gy)) * (float)stride; }
        else if (ch < 4) { vt = vi = expf(acc) * (float)stride; }
        else { vi = sigmoidf_(acc); }
        const long o = ((long)b * A + a0 + p) * nch + ch;
        if (out_train) out_train[o] = vt;
        if (out_infer) out_infer[o] = vi;
    }
}

// backward of the prediction convs: d_raw [B, A, 5+nc] (gradient wrt the raw conv outputs) ->
// d_cls_feat / d_reg_feat [B*hw, Hd] (overwritten) and dW/db (+=).
__global__ __launch_bounds__(256) void head_pred_bwd_feat_kernel(const float* __restrict__ d_raw, const float* __restrict__ cls_w,
                                                                 const float* __restrict__ reg_w, const float* __restrict__ obj_w,
                                                                 float* __restrict__ d_cls_feat, float* __restrict__ d_reg_feat,
                                                                 const float* __restrict__ gscale,
                                                                 int B, int hw, int Hd, int nc, int a0, int A) {
    const int nch = 5 + nc;
    const float gsc = gscale ? gscale[0] : 1.f;
    const long total = (long)B * hw * Hd;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % Hd);
        const long pos = idx / Hd;
        const int p = (int)(pos % hw), b = (int)(pos / hw);
        const float* d = d_raw + ((long)b * A + a0 + p) * nch;
        float r = 0.f, c = 0.f;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) r = fmaf(d[ch], reg_w[(long)ch * Hd + k], r);
        r = fmaf(d[4], obj_w[k], r);
        for (int ch = 0; ch < nc; ++ch) c = fmaf(d[5 + ch], cls_w[(long)ch * Hd + k], c);
        d_reg_feat[idx] = r * gsc;
        d_cls_feat[idx] = c * gsc;
    }
}
// dW[ch][k] += sum_pos d_raw[pos][ch] * feat[pos][k], db[ch] += sum_pos d_raw[pos][ch].
// One workgroup per (tower, 64-column chunk of the features, position slice): blockIdx.x = 0 is the regression tower (channels 0-4:
// box + objectness, reg_feat), 1.. the classification tower (channels 5 + 8 j .., cls_feat) -- a feature value is loaded ONCE for all
// channels of its tower, eight positions per thread in flight (the first version ran one workgroup per channel with one dependent
// load per loop iteration: 110 us for the 40960 positions of level 0).
__global__ __launch_bounds__(256) void head_pred_bwd_w_kernel(const float* __restrict__ d_raw, const float* __restrict__ cls_feat,
                                                              const float* __restrict__ reg_feat, float* __restrict__ d_cls_w,
                                                              float* __restrict__ d_cls_b, float* __restrict__ d_reg_w,
                                                              float* __restrict__ d_reg_b, float* __restrict__ d_obj_w,
                                                              float* __restrict__ d_obj_b, const float* __restrict__ gscale,
                                                              int B, int hw, int Hd, int nc, int a0, int A) {
    constexpr int CH = 8, U = 8;
    __shared__ float red[256];
    const float gsc = gscale ? gscale[0] : 1.f;
    const int nch = 5 + nc;
    const int ch0 = blockIdx.x == 0 ? 0 : 5 + 8 * ((int)blockIdx.x - 1);
    const int nme = blockIdx.x == 0 ? 5 : min(CH, nch - ch0);            // channels of this workgroup
    const int k = blockIdx.y * 64 + (threadIdx.x & 63);
    const int kk = min(k, Hd - 1);
    const int slice = threadIdx.x >> 6;                 // 4 position slices
    const float* feat = blockIdx.x == 0 ? reg_feat : cls_feat;
    float acc[CH], bacc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { acc[c] = 0.f; bacc[c] = 0.f; }
    const long npos = (long)B * hw, pstride = (long)gridDim.z * 4;
    for (long pos0 = (long)blockIdx.z * 4 + slice; pos0 < npos; pos0 += pstride * U) {
        float f[U]; const float* dp[U]; bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pos = pos0 + pstride * u;
            ok[u] = pos < npos;
            const long pc = ok[u] ? pos : npos - 1;
            const int p = (int)(pc % hw), b = (int)(pc / hw);
            f[u] = feat[pc * Hd + kk];
            dp[u] = d_raw + ((long)b * A + a0 + p) * nch + ch0;
        }
        // all U x CH gradient values first (clamped channel: always inside the row), then the arithmetic: loaded inside the channel
        // loop behind `ok[u] ? .. : 0`, every one of the 64 loads of an iteration was waited for on its own
        float dv[U][CH];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) dv[u][c] = dp[u][min(c, nme - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float d = (ok[u] && c < nme) ? dv[u][c] * gsc : 0.f;
                acc[c] = fmaf(d, f[u], acc[c]);
                bacc[c] += d;
            }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (c >= nme) break;                                             // workgroup-uniform
        const int ch = ch0 + c;
        red[threadIdx.x] = acc[c];
        __syncthreads();
        if (slice == 0 && k < Hd) {
            const float v = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
            float* dw = ch < 4 ? d_reg_w + (long)ch * Hd : (ch == 4 ? d_obj_w : d_cls_w + (long)(ch - 5) * Hd);
            atomicAdd(dw + k, v);
        }
        __syncthreads();
        if (blockIdx.y == 0) {
            red[threadIdx.x] = (threadIdx.x & 63) == 0 ? bacc[c] : 0.f;
            __syncthreads();
            if (threadIdx.x == 0) {
                float* db = ch < 4 ? d_reg_b + ch : (ch == 4 ? d_obj_b : d_cls_b + (ch - 5));
                atomicAdd(db, red[0] + red[64] + red[128] + red[192]);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// SimOTA assignment: one workgroup per image.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool in_center(float gcx, float gcy, float gx, float gy, float gs) {
    // yolo_head.py:713-730
    const float xc = (gx + 0.5f) * gs, yc = (gy + 0.5f) * gs;
    const float dist = gs * 1.5f;
    const float l = gcx - dist, r = gcx + dist, t = gcy - dist, b = gcy + dist;
    const float cl = xc - l, cr = r - xc, ct = yc - t, cb = b - yc;
    return fminf(fminf(cl, ct), fminf(cr, cb)) > 0.0f;
}
__device__ __forceinline__ float iou_cxcywh(const float* a, const float* b) {
    // boxes.py:99-113 (xyxy=False): a = gt, b = prediction
    const float tlx = fmaxf(a[0] - a[2] / 2, b[0] - b[2] / 2), tly = fmaxf(a[1] - a[3] / 2, b[1] - b[3] / 2);
    const float brx = fminf(a[0] + a[2] / 2, b[0] + b[2] / 2), bry = fminf(a[1] + a[3] / 2, b[1] + b[3] / 2);
    const float area_a = a[2] * a[3], area_b = b[2] * b[3];
    const float en = (tlx < brx ? 1.f : 0.f) * (tly < bry ? 1.f : 0.f);
    const float area_i = (brx - tlx) * (bry - tly) * en;
    return area_i / (area_a + area_b - area_i);
}
__device__ __forceinline__ float bce_clamped(float p, float t) {
    // F.binary_cross_entropy: log terms clamped at -100
    const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
    return -(t * lp + (1.f - t) * lq);
}

struct AssignOut {
    unsigned char* fg_mask;      // [B, A]
    unsigned char* ignore_mask;  // [B, A]
    int* matched_row;            // [B, A] label row of the matched gt (-1 = none)
    int* matched_valid_idx;      // [B, A] index among the valid gts (reference's matched_gt_inds)
    float* pred_iou;             // [B, A]
    int* num_fg_img;             // [B]
    int* totals;                 // [0] = sum num_fg, [1] = sum num_gt, [2] = status flags
};

#define MAXGT 128
__global__ __launch_bounds__(256) void simota_kernel(const float* __restrict__ outputs, const float* __restrict__ labels,
                                                     float* __restrict__ ws, int* __restrict__ gws, AssignOut o, Levels L,
                                                     int Nmax, int nc, float ignore_label, int cap, int cand_in_lds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int A = L.A;
    const int b = blockIdx.x;
    const int nch = 5 + nc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // candidates lie within 1.5 strides of a gt centre: at most 3 x 3 anchors per gt and level, so `cap` (host: 16 per gt and
    // level, bounded by A) never binds; an overflow would be reported through status bit 2 instead of corrupting LDS
    // The candidate arrays live in LDS when they fit (every shipped configuration) and in the per-image slice of the
    // global workspace otherwise; likewise the per-gt arrays above MAXGT label rows per frame -- the reference has no limit
    // on either (yolo_head.py:606-700), a crowded 1 Mpx frame must not abort a training run.
    int* gimg = gws + (long)b * (3L * cap + 7L * Nmax);
    int* cand = cand_in_lds ? reinterpret_cast<int*>(smem) : gimg;   // [cap] compacted candidate anchors
    int* cnt = cand + cap;                                  // [cap] per-candidate match count
    int* selg = cnt + cap;                                  // [cap] the gt that selected the candidate
    __shared__ float gtb_s[MAXGT][4];
    __shared__ int gtrow_s[MAXGT], gtcls_s[MAXGT], kg_s[MAXGT];
    const bool gt_in_lds = Nmax <= MAXGT;
    float (*gtb)[4] = gt_in_lds ? gtb_s : reinterpret_cast<float (*)[4]>(gimg + 3L * cap);
    int* gtrow = gt_in_lds ? gtrow_s : gimg + 3L * cap + 4L * Nmax;
    int* gtcls = gt_in_lds ? gtcls_s : gtrow + Nmax;
    int* kg = gt_in_lds ? kg_s : gtcls + Nmax;
    __shared__ int s_nw, s_n, s_nvalid, s_npos, s_any_invalid, s_scan[5], s_nfg;
    const float* lab = labels + (long)b * Nmax * 7;
    const float* outb = outputs + (long)b * A * nch;
    float* ws_cost = ws + (long)b * 2 * Nmax * A;
    float* ws_iou = ws_cost + (long)Nmax * A;

    if (tid == 0) {
        int nw = 0, n = 0;
        for (int r = 0; r < Nmax; ++r) {
            float s = 0.f;
            for (int k = 0; k < 7; ++k) s += lab[r * 7 + k];
            const bool nz = s > 0.f, valid = lab[r * 7] != ignore_label;
            nw += nz; n += (nz && valid);
        }
        int nv = 0, anyinv = 0;
        for (int r = 0; r < nw; ++r) {
            if (lab[r * 7] != ignore_label) {
                gtrow[nv] = r; gtcls[nv] = (int)lab[r * 7];
                gtb[nv][0] = lab[r * 7 + 1]; gtb[nv][1] = lab[r * 7 + 2]; gtb[nv][2] = lab[r * 7 + 3]; gtb[nv][3] = lab[r * 7 + 4];
                ++nv;
            } else anyinv = 1;
        }
        s_nw = nw; s_n = n; s_nvalid = nv; s_any_invalid = anyinv; s_npos = 0; s_nfg = 0;
    }
    __syncthreads();
    const int nw = s_nw, nvalid = s_nvalid, n = s_n;
    // ---- geometry: candidate / ignore masks, ordered compaction of candidates -----------------------
    for (int base = 0; base < A; base += 256) {
        const int a = base + tid;
        bool c_all = false, c_valid = false;
        if (a < A) {
            float gx, gy, gs; anchor_geom(L, a, gx, gy, gs);
            for (int r = 0; r < nw; ++r) {
                const bool inc = in_center(lab[r * 7 + 1], lab[r * 7 + 2], gx, gy, gs);
                c_all |= inc;
                if (lab[r * 7] != ignore_label) c_valid |= inc;
            }
            // n == 0 with only-ignore boxes (:832-836): geometry of the first `num_ignore` rows
            if (n == 0) {
                int nign = 0;
                for (int r = 0; r < Nmax; ++r) nign += lab[r * 7] == ignore_label;
                bool ig = false;
                float tot = 0.f;
                for (int r = 0; r < Nmax * 7; ++r) tot += lab[r];
                if (tot != 0.f) for (int r = 0; r < nign; ++r) ig |= in_center(lab[r * 7 + 1], lab[r * 7 + 2], gx, gy, gs);
                o.ignore_mask[(long)b * A + a] = ig;
                c_valid = false;
            } else {
                o.ignore_mask[(long)b * A + a] = s_any_invalid ? (c_all && !c_valid) : 0;
            }
            o.fg_mask[(long)b * A + a] = 0;
            o.matched_row[(long)b * A + a] = -1;
            o.matched_valid_idx[(long)b * A + a] = -1;
            o.pred_iou[(long)b * A + a] = 0.f;
        }
        const bool isc = a < A && c_valid && n > 0;
        const unsigned long long bal = __ballot(isc);
        if (lane == 0) s_scan[wave] = __popcll(bal);
        __syncthreads();
        int off = s_npos;
        for (int w2 = 0; w2 < wave; ++w2) off += s_scan[w2];
        if (isc) { const int j = off + __popcll(bal & ((1ull << lane) - 1)); if (j < cap) { cand[j] = a; cnt[j] = 0; selg[j] = -1; } }
        __syncthreads();
        if (tid == 0) s_npos += s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
        __syncthreads();
    }
    if (s_npos > cap) { if (tid == 0) { o.num_fg_img[b] = 0; atomicOr(o.totals + 2, 2); } return; }
    const int npos = s_npos;
    if (n == 0) { if (tid == 0) { o.num_fg_img[b] = 0; } return; }
    if (tid == 0) atomicAdd(o.totals + 1, n);
    if (npos == 0) {   // reference raises "selected index k out of range" (:744-751)
        if (tid == 0) { o.num_fg_img[b] = 0; atomicOr(o.totals + 2, 1); }
        return;
    }
    // ---- pairwise IoU and cost (:641-675) --------------------------------------------------------
    for (int e = tid; e < nvalid * npos; e += 256) {
        const int g = e / npos, j = e - g * npos;
        const int a = cand[j];
        const float* pr = outb + (long)a * nch;
        const float iou = iou_cxcywh(gtb[g], pr);
        const float iou_loss = -logf(iou + 1e-8f);
        const float so = sigmoidf_(pr[4]);
        float cls_loss = 0.f;
        for (int c = 0; c < nc; ++c) {
            const float p = sqrtf(sigmoidf_(pr[5 + c]) * so);
            cls_loss += bce_clamped(p, c == gtcls[g] ? 1.f : 0.f);
        }
        float gx, gy, gs; anchor_geom(L, a, gx, gy, gs);
        const bool geom = in_center(gtb[g][0], gtb[g][1], gx, gy, gs);
        const float cost = cls_loss + 3.0f * iou_loss + 1e6f * (geom ? 0.f : 1.f);
        ws_cost[(long)g * A + j] = cost;
        ws_iou[(long)g * A + j] = iou;
    }
    __syncthreads();
    // ---- dynamic k (:741-743) and top-k smallest cost per gt (:744-751): one wave per gt -----------
    for (int g = wave; g < nvalid; g += 4) {
        const float* ci = ws_iou + (long)g * A;
        const float* cc = ws_cost + (long)g * A;
        const int kc = min(10, npos);
        float lastv = INFINITY; int lasti = -1; float sum = 0.f;
        for (int it = 0; it < kc; ++it) {          // it-th largest IoU; ties -> smaller index first
            float bv = -INFINITY; int bi = 0x7fffffff;
            for (int j = lane; j < npos; j += 64) {
                const float v = ci[j];
                const bool after = (v < lastv) || (v == lastv && j > lasti);
                if (after && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off, 64); const int oi = __shfl_xor(bi, off, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            sum += bv; lastv = bv; lasti = bi;
        }
        const int k = max((int)sum, 1);
        if (lane == 0) kg[g] = k;
        lastv = -INFINITY; lasti = -1;
        for (int it = 0; it < k && it < npos; ++it) {   // it-th smallest cost; ties -> smaller index first
            float bv = INFINITY; int bi = 0x7fffffff;
            for (int j = lane; j < npos; j += 64) {
                const float v = cc[j];
                const bool after = (v > lastv) || (v == lastv && j > lasti);
                if (after && (v < bv || (v == bv && j < bi))) { bv = v; bi = j; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off, 64); const int oi = __shfl_xor(bi, off, 64);
                if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0 && bi != 0x7fffffff) { atomicAdd(&cnt[bi], 1); selg[bi] = g; }
            lastv = bv; lasti = bi;
        }
    }
    __syncthreads();
    // ---- resolve anchors matched to several gts (:755-760), emit ---------------------------------
    int local_fg = 0;
    for (int j = tid; j < npos; j += 256) {
        const int c = cnt[j];
        if (c == 0) continue;
        int g = selg[j];
        if (c > 1) {
            float bv = INFINITY; g = 0;
            for (int gg = 0; gg < nvalid; ++gg) { const float v = ws_cost[(long)gg * A + j]; if (v < bv) { bv = v; g = gg; } }
        }
        const long oa = (long)b * A + cand[j];
        o.fg_mask[oa] = 1; o.matched_row[oa] = gtrow[g]; o.matched_valid_idx[oa] = g; o.pred_iou[oa] = ws_iou[(long)g * A + j];
        ++local_fg;
    }
    if (local_fg) atomicAdd(&s_nfg, local_fg);
    __syncthreads();
    if (tid == 0) { o.num_fg_img[b] = s_nfg; atomicAdd(o.totals, s_nfg); }
}

// ---------------------------------------------------------------------------------------------------
// losses + gradient wrt the RAW conv outputs (decode folded in).  grid-stride over B*A anchors.
//   sums[0] = sum_fg (1 - iou^2), sums[1] = sum obj loss, sums[2] = sum cls loss  (double)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void yolox_loss_kernel(const float* __restrict__ outputs, const float* __restrict__ labels,
                                                         const unsigned char* __restrict__ fg_mask,
                                                         const unsigned char* __restrict__ ignore_mask,
                                                         const int* __restrict__ matched_row, const float* __restrict__ pred_iou,
                                                         const int* __restrict__ totals, double* __restrict__ sums,
                                                         float* __restrict__ d_raw, Levels L, int B, int Nmax, int nc,
                                                         int focal, float reg_w, float obj_w, float cls_w, float gscale,
                                                         const float* __restrict__ label_w, const double* __restrict__ wsum) {
    __shared__ double red[3][4];
    const int nch = 5 + nc, A = L.A;
    const int nfg_raw = totals[0];
    const float num_fg = (float)max(nfg_raw, 1);
    const float inv_fg_mean = nfg_raw > 0 ? 1.f / (float)nfg_raw : 0.f;      // IOUloss reduction='mean'
    // bbox_loss_weighting (:550-553): per-box weights divided by their mean over the batch's foreground anchors
    const float wnorm = label_w ? (float)nfg_raw / (float)wsum[0] : 1.f;
    double s_iou = 0.0, s_obj = 0.0, s_cls = 0.0;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)B * A; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / A), a = (int)(idx - (long)b * A);
        const float* pr = outputs + idx * nch;
        float* dr = d_raw ? d_raw + idx * nch : nullptr;
        const bool fg = fg_mask[idx] != 0;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
        if (fg) {
            const float* gt = labels + ((long)b * Nmax + matched_row[idx]) * 7 + 1;
            const float px = pr[0], py = pr[1], pw = pr[2], ph = pr[3];
            const float p_l = px - pw / 2, p_r = px + pw / 2, p_t = py - ph / 2, p_b = py + ph / 2;
            const float t_l = gt[0] - gt[2] / 2, t_r = gt[0] + gt[2] / 2, t_t = gt[1] - gt[3] / 2, t_b = gt[1] + gt[3] / 2;
            const float tlx = fmaxf(p_l, t_l), tly = fmaxf(p_t, t_t), brx = fminf(p_r, t_r), bry = fminf(p_b, t_b);
            const float area_p = pw * ph, area_g = gt[2] * gt[3];
            const float en = (tlx < brx ? 1.f : 0.f) * (tly < bry ? 1.f : 0.f);
            const float iw = brx - tlx, ih = bry - tly;
            const float area_i = iw * ih * en;
            const float u = area_p + area_g - area_i + 1e-16f;
            const float iou = area_i / u;
            const float bw = label_w ? label_w[(long)b * Nmax + matched_row[idx]] * wnorm : 1.f;
            s_iou += (double)((1.f - iou * iou) * bw);
            // d(1 - iou^2)
            const float dl_diou = -2.f * iou;
            const float diou_dI = (u + area_i) / (u * u), diou_dP = -area_i / (u * u);
            const float dI_dtlx = -ih * en, dI_dbrx = ih * en, dI_dtly = -iw * en, dI_dbry = iw * en;
            const float al = p_l > t_l ? 1.f : 0.f, ar = p_r < t_r ? 1.f : 0.f, at = p_t > t_t ? 1.f : 0.f, ab = p_b < t_b ? 1.f : 0.f;
            const float dI_dpx = dI_dtlx * al + dI_dbrx * ar, dI_dpy = dI_dtly * at + dI_dbry * ab;
            const float dI_dpw = -0.5f * dI_dtlx * al + 0.5f * dI_dbrx * ar, dI_dph = -0.5f * dI_dtly * at + 0.5f * dI_dbry * ab;
            const float k = reg_w * inv_fg_mean * dl_diou * gscale * bw;
            g0 = k * diou_dI * dI_dpx; g1 = k * diou_dI * dI_dpy;
            g2 = k * (diou_dI * dI_dpw + diou_dP * ph); g3 = k * (diou_dI * dI_dph + diou_dP * pw);
            // class BCE-with-logits against onehot * matched IoU (:507-509)
            const int gcls = (int)labels[((long)b * Nmax + matched_row[idx]) * 7];
            const float piou = pred_iou[idx];
            for (int c = 0; c < nc; ++c) {
                const float x = pr[5 + c], t = c == gcls ? piou : 0.f;
                s_cls += (double)((fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)))) * bw);
                if (dr) dr[5 + c] = (sigmoidf_(x) - t) * cls_w / num_fg * gscale * bw;
            }
        } else if (dr) {
            for (int c = 0; c < nc; ++c) dr[5 + c] = 0.f;
        }
        // objectness on every non-ignored anchor (:564-567 / :930-935)
        float dobj = 0.f;
        if (!ignore_mask[idx]) {
            const float x = pr[4], t = fg ? 1.f : 0.f;
            const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
            if (!focal) { s_obj += (double)ce; dobj = sigmoidf_(x) - t; }
            else {      // torchvision sigmoid_focal_loss(alpha .25, gamma 2)
                const float p = sigmoidf_(x);
                const float p_t = p * t + (1.f - p) * (1.f - t);
                const float a_t = 0.25f * t + 0.75f * (1.f - t);
                const float om = 1.f - p_t;
                s_obj += (double)(a_t * ce * om * om);
                const float dce = p - t;                       // d ce / dx
                const float dpt = (2.f * t - 1.f) * p * (1.f - p);
                dobj = a_t * (dce * om * om - 2.f * ce * om * dpt);
            }
            dobj = dobj * obj_w / num_fg * gscale;
        }
        if (dr) {
            float gx, gy, gs; anchor_geom(L, a, gx, gy, gs);
            dr[0] = g0 * gs; dr[1] = g1 * gs; dr[2] = g2 * pr[2]; dr[3] = g3 * pr[3];     // decode backward
            dr[4] = dobj;
        }
    }
    // block reduce -> 3 atomics
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v[3] = {s_iou, s_obj, s_cls};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
        if (lane == 0) red[k][wave] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(sums + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// sum of the per-box weights over the foreground anchors (the mean that normalises them, :552)
__global__ __launch_bounds__(256) void bbox_wsum_kernel(const unsigned char* __restrict__ fg_mask, const int* __restrict__ matched_row,
                                                        const float* __restrict__ label_w, double* __restrict__ wsum, long BA, int A, int Nmax) {
    __shared__ double red[4];
    double s = 0.0;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < BA; idx += (long)gridDim.x * blockDim.x)
        if (fg_mask[idx]) s += (double)label_w[(idx / A) * Nmax + matched_row[idx]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { const double t = red[0] + red[1] + red[2] + red[3]; if (t != 0.0) atomicAdd(wsum, t); }
}

// ignore_bg_k (_get_highest_score_mask, :335-356): the n = int(#background anchors * k) highest objectness LOGITS among the anchors
// SimOTA left in the background are dropped from the objectness loss (marked in ignore_mask).  One workgroup per image: radix select
// of the n-th largest key (4 x 8 bits), then one marking pass; ties at the threshold go to the lowest anchor indices (torch.topk leaves
// the order among equal scores unspecified).  The step belongs to get_losses only (:541-542): a batch with any ignore box goes
// through get_losses_w_ignore, which has no such step -- every workgroup scans the labels and leaves if it finds one.
__device__ __forceinline__ unsigned f32_order_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(1024) void bg_topk_ignore_kernel(const float* __restrict__ outputs, const float* __restrict__ labels,
                                                              const unsigned char* __restrict__ fg_mask, unsigned char* __restrict__ ignore_mask,
                                                              int BN, int A, int nch, double kfrac, float ignore_label) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_cnt, s_prefix, s_remain, s_flag;
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) { s_cnt = 0; s_flag = 0; }
    __syncthreads();
    for (int j = tid; j < BN; j += blockDim.x)
        if (labels[(long)j * 7] == ignore_label) s_flag = 1;
    const unsigned char* fg = fg_mask + (long)b * A;
    const float* sc = outputs + (long)b * A * nch + 4;
    unsigned local = 0;
    for (int a = tid; a < A; a += blockDim.x) local += fg[a] ? 0u : 1u;
    if (local) atomicAdd(&s_cnt, local);
    __syncthreads();
    if (s_flag) return;
    const int n = (int)((double)(float)s_cnt * kfrac);
    if (n <= 0) return;
    if (tid == 0) { s_prefix = 0; s_remain = (unsigned)n; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix, himask = pass ? 0xFFFFFFFFu << (shift + 8) : 0u;
        for (int a = tid; a < A; a += blockDim.x) {
            if (fg[a]) continue;
            const unsigned key = f32_order_key(sc[(long)a * nch]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned remain = s_remain;
            int d = 255;
            for (; d > 0; --d) { if (hist[d] >= remain) break; remain -= hist[d]; }
            s_prefix = prefix | ((unsigned)d << shift);
            s_remain = remain;                                  // elements still to take among those sharing the new prefix
        }
        __syncthreads();
    }
    const unsigned thr = s_prefix, take_eq = s_remain;
    // hist[thr & 255] of the last pass = number of background anchors whose key equals the threshold
    const bool all_eq = hist[thr & 255u] == take_eq;
    unsigned char* ig = ignore_mask + (long)b * A;
    for (int a = tid; a < A; a += blockDim.x) {
        if (fg[a]) continue;
        const unsigned key = f32_order_key(sc[(long)a * nch]);
        if (key > thr || (all_eq && key == thr)) ig[a] = 1;
    }
    if (!all_eq && tid == 0) {
        unsigned left = take_eq;
        for (int a = 0; a < A && left; ++a)
            if (!fg[a] && f32_order_key(sc[(long)a * nch]) == thr) { ig[a] = 1; --left; }
    }
}

__global__ void yolox_loss_finalize_kernel(const double* __restrict__ sums, const int* __restrict__ totals, float* __restrict__ losses,
                                           float reg_w, float obj_w, float cls_w) {
    const int nfg_raw = totals[0], ngt = totals[1];
    const float num_fg = (float)max(nfg_raw, 1);
    const float li = nfg_raw > 0 ? reg_w * (float)(sums[0] / (double)nfg_raw) : 0.f;
    const float lo = obj_w * ((float)sums[1] / num_fg);
    const float lc = cls_w * ((float)sums[2] / num_fg);
    losses[0] = li + lo + lc; losses[1] = li; losses[2] = lo; losses[3] = lc; losses[4] = 0.f;
    losses[5] = num_fg / (float)max(ngt, 1);
}

// ---------------------------------------------------------------------------------------------------
// postprocess + batched NMS: one workgroup (1024 threads) per image.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool iou_gt(const float* a, const float* b, float thr) {
    // areas are recomputed from the (offset) boxes exactly as torchvision precomputes them: (x2-x1)*(y2-y1)
    const float area_a = (a[2] - a[0]) * (a[3] - a[1]), area_b = (b[2] - b[0]) * (b[3] - b[1]);
    const float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]), xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    return inter / (area_a + area_b - inter) > thr;
}

// pred [B,A,5+nc] (cx,cy,w,h,obj,cls..): boxes are rewritten IN PLACE to xyxy like boxes.py:41-46.
// det_out [B, max_det, 7] = (x1,y1,x2,y2,obj,cls_conf,cls_id) in NMS order ; det_cnt[B]
__global__ __launch_bounds__(1024) void postprocess_nms_kernel(float* __restrict__ pred, float* __restrict__ det_out,
                                                               int* __restrict__ det_cnt, int A, int nc, int ncols,
                                                               float conf_thre, float nms_thre, int class_agnostic,
                                                               int max_det, int vanilla_limit, int convert_boxes, int cap,
                                                               char* __restrict__ gws, long gws_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = 16;
    // cap = most candidates (score >= conf_thre) the LDS arrays hold: A when everything fits (Gen1 1680, Gen4 5040 anchors),
    // 4096 for larger heads (1 Mpx: 20160 anchors).  An image with more candidates than that re-runs the compaction into
    // its slice of the global workspace (sized for all A anchors) and sorts / suppresses there: no limit, as in the
    // reference (boxes.py:53-80); without a workspace it reports det_cnt = -1.
    int NP2 = 1; while (NP2 < cap) NP2 <<= 1;
    float* skey = reinterpret_cast<float*>(smem);            // [NP2] scores (sorted desc)
    int* sidx = reinterpret_cast<int*>(skey + NP2);          // [NP2] anchor index
    float* sbox = reinterpret_cast<float*>(sidx + NP2);      // [cap][4] boxes in sorted order (with class offset)
    unsigned char* removed = reinterpret_cast<unsigned char*>(sbox + 4 * (size_t)cap);   // [cap]
    __shared__ int s_scan[NW], s_n, s_keep;
    __shared__ float s_red[NW];
    __shared__ unsigned long long s_kept;
    float* pb = pred + (long)b * A * ncols;
    if (tid == 0) { s_n = 0; s_keep = 0; }
    __syncthreads();
    // 1. xywh -> xyxy in place, class max, confidence mask, ordered compaction
    auto compact = [&](bool convert, int limit) {
        for (int base = 0; base < A; base += 1024) {
            const int a = base + tid;
            bool ok = false; float score = 0.f;
            if (a < A) {
                float* p = pb + (long)a * ncols;
                if (convert) {
                    const float cx = p[0], cy = p[1], w = p[2], h = p[3];
                    p[0] = cx - w / 2; p[1] = cy - h / 2; p[2] = cx + w / 2; p[3] = cy + h / 2;
                }
                float cc;
                if (nc > 0) { cc = p[5]; for (int c = 1; c < nc; ++c) cc = fmaxf(cc, p[5 + c]); } else cc = p[5];
                score = p[4] * cc;
                ok = score >= conf_thre;
            }
            const unsigned long long bal = __ballot(ok);
            if (lane == 0) s_scan[wave] = __popcll(bal);
            __syncthreads();
            int off = s_n;
            for (int w2 = 0; w2 < wave; ++w2) off += s_scan[w2];
            if (ok) { const int j = off + __popcll(bal & ((1ull << lane) - 1)); if (j < limit) { skey[j] = score; sidx[j] = a; } }
            __syncthreads();
            if (tid == 0) { int t = 0; for (int w2 = 0; w2 < NW; ++w2) t += s_scan[w2]; s_n += t; }
            __syncthreads();
        }
    };
    compact(convert_boxes != 0, cap);
    const int n = s_n;
    if (n > cap) {                                       // boxes were converted in place like the reference does
        if (!gws) { if (tid == 0) det_cnt[b] = -1; return; }
        int AP2 = 1; while (AP2 < A) AP2 <<= 1;
        char* g = gws + (long)b * gws_stride;
        skey = reinterpret_cast<float*>(g);
        sidx = reinterpret_cast<int*>(skey + AP2);
        sbox = reinterpret_cast<float*>(sidx + AP2);
        removed = reinterpret_cast<unsigned char*>(sbox + 4 * (size_t)A);
        __syncthreads();
        if (tid == 0) s_n = 0;
        __syncthreads();
        compact(false, A);
    }
    if (n == 0) { if (tid == 0) det_cnt[b] = 0; return; }
    int np2 = 1; while (np2 < n) np2 <<= 1;
    for (int j = n + tid; j < np2; j += 1024) { skey[j] = -INFINITY; sidx[j] = 0x7fffffff; }
    __syncthreads();
    // 2. bitonic sort: score descending, ties by ascending original position (= stable sort)
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < np2; t += 1024) {
                const int ixj = t ^ j;
                if (ixj > t) {
                    const float ka = skey[t], kb = skey[ixj]; const int ia = sidx[t], ib = sidx[ixj];
                    const bool a_first = (ka > kb) || (ka == kb && ia < ib);     // a should precede b
                    const bool up = (t & k) == 0;
                    if (up ? !a_first : a_first) { skey[t] = kb; skey[ixj] = ka; sidx[t] = ib; sidx[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    // 3. class offsets (coordinate trick) unless class-agnostic or above the per-class ("vanilla") limit
    const bool vanilla = !class_agnostic && (4 * n > vanilla_limit);
    float maxc = -INFINITY;
    if (!class_agnostic && !vanilla) {
        for (int j = tid; j < n; j += 1024) {
            const float* p = pb + (long)sidx[j] * ncols;
            maxc = fmaxf(fmaxf(maxc, fmaxf(p[0], p[1])), fmaxf(p[2], p[3]));
        }
        maxc = wave_max(maxc);
        if (lane == 0) s_red[wave] = maxc;
        __syncthreads();
        maxc = s_red[0];
        for (int w2 = 1; w2 < NW; ++w2) maxc = fmaxf(maxc, s_red[w2]);
    }
    const float offs_unit = maxc + 1.0f;
    for (int j = tid; j < n; j += 1024) {
        const float* p = pb + (long)sidx[j] * ncols;
        int cid = 0;
        if (nc > 0) { float cc = p[5]; for (int c = 1; c < nc; ++c) if (p[5 + c] > cc) { cc = p[5 + c]; cid = c; } }
        else cid = (int)p[6];
        float off = 0.f;
        if (!class_agnostic && !vanilla) off = (float)cid * offs_unit;
        const float x1 = p[0] + off, y1 = p[1] + off, x2 = p[2] + off, y2 = p[3] + off;
        sbox[4 * j] = x1; sbox[4 * j + 1] = y1; sbox[4 * j + 2] = x2; sbox[4 * j + 3] = y2;
        removed[j] = 0;
    }
    __syncthreads();
    // 4. greedy suppression in chunks of 64 sorted boxes
    for (int s0 = 0; s0 < n; s0 += 64) {
        if (wave == 0) {
            const int j = s0 + lane;
            const bool valid = j < n;
            int myc = 0;
            if (vanilla && valid) { const float* p = pb + (long)sidx[j] * ncols; float cc = p[5]; for (int c = 1; c < nc; ++c) if (p[5 + c] > cc) { cc = p[5 + c]; myc = c; } }
            unsigned long long m = 0;
            if (valid) {
                for (int t = lane + 1; t < 64 && s0 + t < n; ++t) {
                    bool sup = iou_gt(sbox + 4 * j, sbox + 4 * (s0 + t), nms_thre);
                    if (sup && vanilla) {
                        const float* p = pb + (long)sidx[s0 + t] * ncols; int oc = 0; float cc = p[5];
                        for (int c = 1; c < nc; ++c) if (p[5 + c] > cc) { cc = p[5 + c]; oc = c; }
                        sup = oc == myc;
                    }
                    if (sup) m |= 1ull << t;
                }
            }
            unsigned long long alive = __ballot(valid && !removed[j]);
            for (int t = 0; t < 64; ++t) {
                const unsigned long long mt = __shfl(m, t, 64);
                if (alive & (1ull << t)) alive &= ~mt;
            }
            if (lane == 0) s_kept = alive;
            if (valid && !(alive & (1ull << lane))) removed[j] = 1;
        }
        __syncthreads();
        const unsigned long long kept = s_kept;
        if (kept) {
            for (int j = s0 + 64 + tid; j < n; j += 1024) {
                if (removed[j]) continue;
                int myc = 0;
                if (vanilla) { const float* p = pb + (long)sidx[j] * ncols; float cc = p[5]; for (int c = 1; c < nc; ++c) if (p[5 + c] > cc) { cc = p[5 + c]; myc = c; } }
                unsigned long long kk = kept;
                while (kk) {
                    const int t = __ffsll((long long)kk) - 1; kk &= kk - 1;
                    bool sup = iou_gt(sbox + 4 * (s0 + t), sbox + 4 * j, nms_thre);
                    if (sup && vanilla) {
                        const float* p = pb + (long)sidx[s0 + t] * ncols; int oc = 0; float cc = p[5];
                        for (int c = 1; c < nc; ++c) if (p[5 + c] > cc) { cc = p[5 + c]; oc = c; }
                        sup = oc == myc;
                    }
                    if (sup) { removed[j] = 1; break; }
                }
            }
        }
        __syncthreads();
    }
    // 5. ordered emit
    for (int base = 0; base < n; base += 1024) {
        const int j = base + tid;
        const bool ok = j < n && !removed[j];
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) s_scan[wave] = __popcll(bal);
        __syncthreads();
        int off = s_keep;
        for (int w2 = 0; w2 < wave; ++w2) off += s_scan[w2];
        if (ok) {
            const int k = off + __popcll(bal & ((1ull << lane) - 1));
            if (k < max_det) {
                const float* p = pb + (long)sidx[j] * ncols;
                float* d = det_out + ((long)b * max_det + k) * 7;
                float cc; int cid = 0;
                if (nc > 0) { cc = p[5]; for (int c = 1; c < nc; ++c) if (p[5 + c] > cc) { cc = p[5 + c]; cid = c; } }
                else { cc = p[5]; cid = (int)p[6]; }
                d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3]; d[4] = p[4]; d[5] = cc; d[6] = (float)cid;
            }
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w2 = 0; w2 < NW; ++w2) t += s_scan[w2]; s_keep += t; }
        __syncthreads();
    }
    if (tid == 0) det_cnt[b] = min(s_keep, max_det);
}

// ---------------------------------------------------------------------------------------------------
// pseudo-label filter (ssod.py:40-188): det [B, max_det, 7] + det_cnt -> labels [B, max_det, 8] + lab_cnt
//   (t=0, x, y, w, h, cls_id, cls_conf, obj), corner xy, order preserved.  One wave per image.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pseudo_filter_kernel(const float* __restrict__ det, const int* __restrict__ det_cnt,
                                                           float* __restrict__ lab, int* __restrict__ lab_cnt, int max_det,
                                                           const float* __restrict__ obj_thr, const float* __restrict__ cls_thr,
                                                           int nthr, int filter_boxes, float frame_w, float frame_h) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = det_cnt[b];
    if (n < 0) { if (lane == 0) lab_cnt[b] = -1; return; }          // NMS overflow report travels with the counts
    int outn = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        bool ok = false; float x1 = 0, y1 = 0, x2 = 0, y2 = 0, ob = 0, cc = 0, cid = 0;
        if (j < n) {
            const float* d = det + ((long)b * max_det + j) * 7;
            x1 = d[0]; y1 = d[1]; x2 = d[2]; y2 = d[3]; ob = d[4]; cc = d[5]; cid = d[6];
            bool so = false, sc = false;
            if (nthr == 1) { so = ob > obj_thr[0]; sc = cc > cls_thr[0]; }
            else for (int k = 0; k < nthr; ++k) { if (cid == (float)k) { so |= ob > obj_thr[k]; sc |= cc > cls_thr[k]; } }
            ok = so && sc;
            if (filter_boxes) {
                x1 = fminf(fmaxf(x1, 0.f), frame_w - 1.f); y1 = fminf(fmaxf(y1, 0.f), frame_h - 1.f);
                x2 = fminf(fmaxf(x2, 0.f), frame_w - 1.f); y2 = fminf(fmaxf(y2, 0.f), frame_h - 1.f);
                const float w = x2 - x1, h = y2 - y1;
                const float maxw = (float)((9 * (int)frame_w) / 10);
                ok = ok && (w > 0.f) && (h > 0.f) && (w >= 5.f) && (h >= 5.f) && (w <= maxw);
            }
        }
        const unsigned long long bal = __ballot(ok);
        if (ok) {
            const int k = outn + __popcll(bal & ((1ull << lane) - 1));
            float* l = lab + ((long)b * max_det + k) * 8;
            l[0] = 0.f; l[1] = x1; l[2] = y1; l[3] = x2 - x1; l[4] = y2 - y1; l[5] = cid; l[6] = cc; l[7] = ob;
        }
        outn += __popcll(bal);
    }
    if (lane == 0) lab_cnt[b] = outn;
}

// ===================================================================================================
static Levels make_levels(int nlv, const int* hs, const int* wsz, const int* strides) {
    Levels L{}; L.n = nlv; int a = 0;
    for (int k = 0; k < nlv && k < MAXLVL; ++k) { L.h[k] = hs[k]; L.w[k] = wsz[k]; L.stride[k] = strides[k]; L.a0[k] = a; a += hs[k] * wsz[k]; }
    L.A = a; return L;
}
static inline int flat_grid(long n) { return (int)min((long)4096, max((long)1, (n + 255) / 256)); }

LEOD_API int leod_head_pred_fwd(const float* cls_feat, const float* reg_feat, const float* cls_w, const float* cls_b,
                                const float* reg_w, const float* reg_b, const float* obj_w, const float* obj_b,
                                float* out_train, float* out_infer, int B, int h, int w, int Hd, int nc, int stride, int a0,
                                int A, hipStream_t stream) {
    if (!cls_feat || !reg_feat || (Hd & 3) || (!out_train && !out_infer)) return LEOD_ERR_ARG;
    const long total = (long)B * h * w * (5 + nc);
    if (total == 0) return LEOD_OK;
    static const int lds_on = 1;
    if (lds_on && Hd <= 128 && nc <= 16 && (long)B * h * w >= 4096) {
        const int grid = (int)min((long)2048, ((long)B * h * w + 63) / 64);
        if (Hd <= 96) hipLaunchKernelGGL(head_pred_fwd_lds_kernel<96>, dim3(grid), dim3(256), 0, stream, cls_feat, reg_feat, cls_w, cls_b,
                                         reg_w, reg_b, obj_w, obj_b, out_train, out_infer, B, h * w, w, Hd, nc, stride, a0, A);
        else hipLaunchKernelGGL(head_pred_fwd_lds_kernel<128>, dim3(grid), dim3(256), 0, stream, cls_feat, reg_feat, cls_w, cls_b,
                                reg_w, reg_b, obj_w, obj_b, out_train, out_infer, B, h * w, w, Hd, nc, stride, a0, A);
        return leod_launch_status();
    }
    hipLaunchKernelGGL(head_pred_fwd_kernel, dim3(flat_grid(total)), dim3(256), 0, stream, cls_feat, reg_feat, cls_w, cls_b,
                       reg_w, reg_b, obj_w, obj_b, out_train, out_infer, B, h * w, w, Hd, nc, stride, a0, A);
    return leod_launch_status();
}

LEOD_API int leod_head_pred_bwd(const float* d_raw, const float* cls_feat, const float* reg_feat, const float* cls_w,
                                const float* reg_w, const float* obj_w, float* d_cls_feat, float* d_reg_feat, float* d_cls_w,
                                float* d_cls_b, float* d_reg_w, float* d_reg_b, float* d_obj_w, float* d_obj_b,
                                const float* gscale, int B, int h, int w, int Hd, int nc, int a0, int A, hipStream_t stream) {
    if (!d_raw || !cls_feat || !reg_feat || !d_cls_feat || !d_reg_feat) return LEOD_ERR_ARG;
    const long total = (long)B * h * w * Hd;
    if (total == 0) return LEOD_OK;
    hipLaunchKernelGGL(head_pred_bwd_feat_kernel, dim3(flat_grid(total)), dim3(256), 0, stream, d_raw, cls_w, reg_w, obj_w,
                       d_cls_feat, d_reg_feat, gscale, B, h * w, Hd, nc, a0, A);
    const int zs = (int)min((long)128, max((long)1, ((long)B * h * w + 127) / 128));      // >= 32 positions per thread
    hipLaunchKernelGGL(head_pred_bwd_w_kernel, dim3(1 + cdiv(nc, 8), cdiv(Hd, 64), zs), dim3(256), 0, stream, d_raw, cls_feat, reg_feat,
                       d_cls_w, d_cls_b, d_reg_w, d_reg_b, d_obj_w, d_obj_b, gscale, B, h * w, Hd, nc, a0, A);
    return leod_launch_status();
}

// candidates lie within 1.5 strides of a gt centre: at most 3 x 3 anchors per gt and level; 16 leaves slack
static inline int simota_cap(int Nmax, int nlv, int A) { return (int)max(1L, min((long)A, 16L * nlv * Nmax)); }

// workspace floats needed by leod_simota_assign: cost + IoU matrices [B][2][Nmax][A], then (as ints) the per-image candidate
// and gt arrays used when they do not fit in LDS (3 * cap + 7 * Nmax per image; cap <= A)
LEOD_API long leod_simota_workspace_floats(int B, int Nmax, int A) { return 2L * B * Nmax * A + (long)B * (3L * A + 7L * Nmax); }

// totals[3] (int, zeroed by the caller): sum num_fg, sum num_gt, status bits (1 = a gt had no candidate anchor)
LEOD_API int leod_simota_assign(const float* outputs, const float* labels, float* workspace, unsigned char* fg_mask,
                                unsigned char* ignore_mask, int* matched_row, int* matched_valid_idx, float* pred_iou,
                                int* num_fg_img, int* totals, int B, int Nmax, int nc, int nlv, const int* hs, const int* wsz,
                                const int* strides, float ignore_label, hipStream_t stream) {
    if (!outputs || !labels || !workspace || !fg_mask || !ignore_mask || !matched_row || !matched_valid_idx || !pred_iou ||
        !num_fg_img || !totals || nlv < 1 || nlv > MAXLVL || Nmax < 1)
        return LEOD_ERR_ARG;
    if (B == 0) return LEOD_OK;
    const Levels L = make_levels(nlv, hs, wsz, strides);
    const int cap = simota_cap(Nmax, nlv, L.A);
    const int cand_in_lds = (size_t)cap * 3 * sizeof(int) <= 120 * 1024;
    const size_t shm = cand_in_lds ? (size_t)cap * 3 * sizeof(int) : 16;
    int* gws = reinterpret_cast<int*>(workspace + 2L * B * Nmax * L.A);
    AssignOut o{fg_mask, ignore_mask, matched_row, matched_valid_idx, pred_iou, num_fg_img, totals};
    static int shm_set = 0;     // raise the dynamic-LDS limit once (not a stream operation; keeps graph capture clean)
    if (shm_set < (int)shm) { (void)hipFuncSetAttribute((const void*)simota_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); shm_set = (int)shm; }
    hipLaunchKernelGGL(simota_kernel, dim3(B), dim3(256), shm, stream, outputs, labels, workspace, gws, o, L, Nmax, nc, ignore_label,
                       cap, cand_in_lds);
    return leod_launch_status();
}

// sums[3] double zeroed by the caller; losses[6] = (loss, iou, obj, cls, l1=0, num_fg/num_gt); d_raw may be NULL
LEOD_API int leod_yolox_loss(const float* outputs, const float* labels, const unsigned char* fg_mask,
                             const unsigned char* ignore_mask, const int* matched_row, const float* pred_iou,
                             const int* totals, double* sums, float* losses, float* d_raw, int B, int Nmax, int nc, int nlv,
                             const int* hs, const int* wsz, const int* strides, int focal, float reg_weight, float obj_weight,
                             float cls_weight, float grad_scale, hipStream_t stream) {
    if (!outputs || !labels || !fg_mask || !ignore_mask || !matched_row || !pred_iou || !totals || !sums || !losses) return LEOD_ERR_ARG;
    const Levels L = make_levels(nlv, hs, wsz, strides);
    if (B > 0)
        hipLaunchKernelGGL(yolox_loss_kernel, dim3(flat_grid((long)B * L.A)), dim3(256), 0, stream, outputs, labels, fg_mask,
                           ignore_mask, matched_row, pred_iou, totals, sums, d_raw, L, B, Nmax, nc, focal, reg_weight,
                           obj_weight, cls_weight, grad_scale, (const float*)nullptr, (const double*)nullptr);
    hipLaunchKernelGGL(yolox_loss_finalize_kernel, dim3(1), dim3(1), 0, stream, sums, totals, losses, reg_weight, obj_weight, cls_weight);
    return leod_launch_status();
}

// leod_yolox_loss with bbox_loss_weighting (yolo_head.py:358-381, :550-553): label_w [B, Nmax] = the configured expression of every
// label row's confidence (the host evaluates it on the label tensor: it is elementwise, so gathering after it equals the reference's
// order); the IoU and class terms (and their gradients) of a foreground anchor are scaled by label_w[its box] / mean over all
// foreground anchors of the batch.  wsum[1] double, caller-zeroed.
LEOD_API int leod_yolox_loss_weighted(const float* outputs, const float* labels, const unsigned char* fg_mask,
                                      const unsigned char* ignore_mask, const int* matched_row, const float* pred_iou,
                                      const int* totals, const float* label_w, double* wsum, double* sums, float* losses, float* d_raw,
                                      int B, int Nmax, int nc, int nlv, const int* hs, const int* wsz, const int* strides, int focal,
                                      float reg_weight, float obj_weight, float cls_weight, float grad_scale, hipStream_t stream) {
    if (!outputs || !labels || !fg_mask || !ignore_mask || !matched_row || !pred_iou || !totals || !sums || !losses || !label_w || !wsum)
        return LEOD_ERR_ARG;
    const Levels L = make_levels(nlv, hs, wsz, strides);
    if (B > 0) {
        hipLaunchKernelGGL(bbox_wsum_kernel, dim3(flat_grid((long)B * L.A)), dim3(256), 0, stream, fg_mask, matched_row, label_w, wsum,
                           (long)B * L.A, L.A, Nmax);
        hipLaunchKernelGGL(yolox_loss_kernel, dim3(flat_grid((long)B * L.A)), dim3(256), 0, stream, outputs, labels, fg_mask,
                           ignore_mask, matched_row, pred_iou, totals, sums, d_raw, L, B, Nmax, nc, focal, reg_weight,
                           obj_weight, cls_weight, grad_scale, label_w, (const double*)wsum);
    }
    hipLaunchKernelGGL(yolox_loss_finalize_kernel, dim3(1), dim3(1), 0, stream, sums, totals, losses, reg_weight, obj_weight, cls_weight);
    return leod_launch_status();
}

// ignore_bg_k (yolo_head.py:335-356, :541-542): outputs [B, A, 5 + nc] (objectness LOGIT in column 4), labels [B, Nmax, 7] after
// _ignore_bbox, fg_mask / ignore_mask [B, A] of leod_simota_assign; marks the top int(#background * k) background logits of every
// image in ignore_mask -- unless any label row carries ignore_label (then the reference's other loss routine runs, without this step).
LEOD_API int leod_bg_topk_ignore(const float* outputs, const float* labels, const unsigned char* fg_mask, unsigned char* ignore_mask,
                                 int B, int Nmax, int A, int nc, double k, float ignore_label, hipStream_t stream) {
    if (!outputs || !labels || !fg_mask || !ignore_mask || A <= 0 || !(k <= 1.0)) return LEOD_ERR_ARG;
    if (B == 0 || !(k > 0.0)) return LEOD_OK;
    hipLaunchKernelGGL(bg_topk_ignore_kernel, dim3(B), dim3(1024), 0, stream, outputs, labels, fg_mask, ignore_mask, B * Nmax, A, 5 + nc,
                       k, ignore_label);
    return leod_launch_status();
}

// postprocess (boxes.py:32-86): nc > 0: pred rows are (cx,cy,w,h,obj,cls_0..cls_nc-1), boxes converted in place.
// nc == 0: rows are already (x1,y1,x2,y2,obj,cls_conf,cls_id) (TTA merge, tta.py:18-61 / pseudo_labeler.py:37-91).
// vanilla_limit: box-element count above which torchvision loops per class (20000 on GPU, 4000 on CPU).
static inline size_t nms_array_bytes(int cap_) { int np2 = 1; while (np2 < cap_) np2 <<= 1; return (size_t)np2 * 8 + (size_t)cap_ * 17 + 16; }

// bytes of the optional global workspace of leod_postprocess_nms: 0 when the candidate arrays of all A anchors fit in LDS
// (Gen1 / Gen4 heads), else one slice per image for the images that exceed the 4096-candidate LDS tier
LEOD_API long leod_postprocess_nms_workspace_bytes(int B, int A) {
    if (A <= 0 || nms_array_bytes(A) <= 156 * 1024) return 0;
    return (long)B * (long)((nms_array_bytes(A) + 255) / 256 * 256);
}

LEOD_API int leod_postprocess_nms(float* pred, float* det_out, int* det_cnt, void* workspace, int B, int A, int nc, float conf_thre,
                                  float nms_thre, int class_agnostic, int max_det, int vanilla_limit, hipStream_t stream) {
    if (!pred || !det_out || !det_cnt || A <= 0) return LEOD_ERR_ARG;
    if (B == 0) return LEOD_OK;
    auto lds_bytes = nms_array_bytes;
    const int cap = lds_bytes(A) <= 156 * 1024 ? A : 4096;
    const long gstride = (long)((nms_array_bytes(A) + 255) / 256 * 256);
    const size_t shm = lds_bytes(cap);
    static int shm_set = 0;
    if (shm_set < (int)shm) { (void)hipFuncSetAttribute((const void*)postprocess_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); shm_set = (int)shm; }
    const int ncols = nc > 0 ? 5 + nc : 7;
    hipLaunchKernelGGL(postprocess_nms_kernel, dim3(B), dim3(1024), shm, stream, pred, det_out, det_cnt, A, nc, ncols, conf_thre,
                       nms_thre, class_agnostic, max_det, vanilla_limit, nc > 0 ? 1 : 0, cap, static_cast<char*>(workspace), gstride);
    return leod_launch_status();
}

LEOD_API int leod_pseudo_filter(const float* det, const int* det_cnt, float* lab, int* lab_cnt, int B, int max_det,
                                const float* obj_thr, const float* cls_thr, int nthr, int filter_boxes, float frame_w,
                                float frame_h, hipStream_t stream) {
    if (!det || !det_cnt || !lab || !lab_cnt || !obj_thr || !cls_thr || nthr < 1) return LEOD_ERR_ARG;
    if (B == 0) return LEOD_OK;
    hipLaunchKernelGGL(pseudo_filter_kernel, dim3(B), dim3(64), 0, stream, det, det_cnt, lab, lab_cnt, max_det, obj_thr, cls_thr,
                       nthr, filter_boxes, frame_w, frame_h);
    return leod_launch_status();
}
