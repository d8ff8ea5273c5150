"""Oracle: YOLOX PAFPN neck + decoupled head + SimOTA assignment + losses, functional
PyTorch-CPU fp32.  TEST INFRASTRUCTURE (see oracle/__init__.py).  Citations relative to
/root/reference.
"""
import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# conv blocks (models/detection/yolox/models/network_blocks.py)
# ------------------------------------------------------------------------------------------------
def base_conv(x, sd, prefix, stride=1, training=False):
    """BaseConv.forward, network_blocks.py:29-51: conv(no bias, pad (k-1)//2) -> BatchNorm2d -> SiLU.
    In training mode batch statistics are used and the running buffers in ``sd`` are updated in
    place (momentum 0.1, unbiased variance), like nn.BatchNorm2d."""
    if (prefix + '.dconv.conv.weight') in sd:       # DWConv, network_blocks.py:57-76: depthwise k x k BaseConv -> pointwise 1 x 1 BaseConv
        return base_conv(base_conv(x, sd, prefix + '.dconv', stride=stride, training=training), sd, prefix + '.pconv', training=training)
    w = sd[prefix + '.conv.weight']
    k = w.shape[-1]
    groups = x.shape[1] if (w.shape[1] == 1 and x.shape[1] > 1 and w.shape[0] == x.shape[1]) else 1    # groups=in_channels (:63)
    y = F.conv2d(x, w, None, stride=stride, padding=(k - 1) // 2, groups=groups)
    rm, rv = sd[prefix + '.bn.running_mean'], sd[prefix + '.bn.running_var']
    if training and (prefix + '.bn.num_batches_tracked') in sd:
        sd[prefix + '.bn.num_batches_tracked'] += 1
    y = F.batch_norm(y, rm, rv, sd[prefix + '.bn.weight'], sd[prefix + '.bn.bias'],
                     training=training, momentum=0.1, eps=1e-5)
    return F.silu(y)


def csp_layer(x, sd, prefix, n, training=False):
    """CSPLayer.forward with shortcut=False, network_blocks.py:104-142 (Bottleneck :79-101, expansion 1.0)."""
    x1 = base_conv(x, sd, prefix + '.conv1', training=training)
    x2 = base_conv(x, sd, prefix + '.conv2', training=training)
    for i in range(n):
        x1 = base_conv(base_conv(x1, sd, f'{prefix}.m.{i}.conv1', training=training),
                       sd, f'{prefix}.m.{i}.conv2', training=training)
    return base_conv(torch.cat((x1, x2), dim=1), sd, prefix + '.conv3', training=training)


def upsample2(x):
    """yolo_pafpn.py:47: interpolate(scale 2, 'nearest-exact')."""
    return F.interpolate(x, scale_factor=2, mode='nearest-exact')


def pafpn_forward(feats: Dict[int, torch.Tensor], sd, n_bottleneck: int, in_stages=(2, 3, 4),
                  prefix='fpn', training=False):
    """YOLOPAFPN.forward, models/detection/yolox_extension/models/yolo_pafpn.py:109-140."""
    x2, x1, x0 = [feats[s] for s in in_stages]
    fpn_out0 = base_conv(x0, sd, prefix + '.lateral_conv0', training=training)
    f_out0 = torch.cat([upsample2(fpn_out0), x1], 1)
    f_out0 = csp_layer(f_out0, sd, prefix + '.C3_p4', n_bottleneck, training)
    fpn_out1 = base_conv(f_out0, sd, prefix + '.reduce_conv1', training=training)
    f_out1 = torch.cat([upsample2(fpn_out1), x2], 1)
    pan_out2 = csp_layer(f_out1, sd, prefix + '.C3_p3', n_bottleneck, training)
    p_out1 = base_conv(pan_out2, sd, prefix + '.bu_conv2', stride=2, training=training)
    p_out1 = torch.cat([p_out1, fpn_out1], 1)
    pan_out1 = csp_layer(p_out1, sd, prefix + '.C3_n3', n_bottleneck, training)
    p_out0 = base_conv(pan_out1, sd, prefix + '.bu_conv1', stride=2, training=training)
    p_out0 = torch.cat([p_out0, fpn_out0], 1)
    pan_out0 = csp_layer(p_out0, sd, prefix + '.C3_n4', n_bottleneck, training)
    return pan_out2, pan_out1, pan_out0


# ------------------------------------------------------------------------------------------------
# head (models/detection/yolox/models/yolo_head.py)
# ------------------------------------------------------------------------------------------------
def head_raw(xin: Sequence[torch.Tensor], sd, prefix='yolox_head', training=False):
    """Per-level raw maps (reg[B,4,h,w], obj[B,1,h,w], cls[B,nc,h,w]) -- yolo_head.py:208-222."""
    outs = []
    for k, x in enumerate(xin):
        x = base_conv(x, sd, f'{prefix}.stems.{k}', training=training)
        cf = base_conv(base_conv(x, sd, f'{prefix}.cls_convs.{k}.0', training=training),
                       sd, f'{prefix}.cls_convs.{k}.1', training=training)
        cls_o = F.conv2d(cf, sd[f'{prefix}.cls_preds.{k}.weight'], sd[f'{prefix}.cls_preds.{k}.bias'])
        rf = base_conv(base_conv(x, sd, f'{prefix}.reg_convs.{k}.0', training=training),
                       sd, f'{prefix}.reg_convs.{k}.1', training=training)
        reg_o = F.conv2d(rf, sd[f'{prefix}.reg_preds.{k}.weight'], sd[f'{prefix}.reg_preds.{k}.bias'])
        obj_o = F.conv2d(rf, sd[f'{prefix}.obj_preds.{k}.weight'], sd[f'{prefix}.obj_preds.{k}.bias'])
        outs.append((reg_o, obj_o, cls_o))
    return outs


def make_grids(hws, strides, dtype=torch.float32):
    """Anchor order: levels 8->16->32, row-major (y,x); grid[...,0]=x (yolo_head.py:297-298,316-326)."""
    xs, ys, ss = [], [], []
    for (h, w), s in zip(hws, strides):
        yv, xv = torch.meshgrid([torch.arange(h, dtype=dtype), torch.arange(w, dtype=dtype)], indexing='ij')
        xs.append(xv.reshape(-1))
        ys.append(yv.reshape(-1))
        ss.append(torch.full((h * w,), float(s), dtype=dtype))
    return torch.cat(xs), torch.cat(ys), torch.cat(ss)


def head_forward(xin, sd, strides, labels=None, prefix='yolox_head', training=False, **loss_kw):
    """YOLOXHead.forward, yolo_head.py:195-287.  Returns (decoded outputs [B,A,5+nc], losses|None).

    Training: the loss is computed on the *logit* obj/cls with decoded boxes (train_outputs);
    the returned tensor is always the inference-style decoded output (sigmoid probabilities)."""
    raw = head_raw(xin, sd, prefix, training)
    hws = [r[0].shape[-2:] for r in raw]
    gx, gy, gs = make_grids(hws, strides, xin[0].dtype)
    B = xin[0].shape[0]
    # [B, A, 5+nc] with logits
    flat = torch.cat([torch.cat([r, o, c], 1).flatten(2) for (r, o, c) in raw], dim=2).permute(0, 2, 1)
    xy = (flat[..., 0:2] + torch.stack([gx, gy], -1)) * gs[:, None]
    wh = torch.exp(flat[..., 2:4]) * gs[:, None]
    losses = None
    if training:
        train_out = torch.cat([xy, wh, flat[..., 4:]], dim=-1)
        losses = get_losses(gx, gy, gs, labels, train_out, **loss_kw)
    out = torch.cat([xy, wh, flat[..., 4:].sigmoid()], dim=-1)
    del B
    return out, losses


# ------------------------------------------------------------------------------------------------
# SimOTA (yolo_head.py:606-774, 974-1148)
# ------------------------------------------------------------------------------------------------
def bboxes_iou_cxcywh(a, b):
    """bboxes_iou(xyxy=False), models/detection/yolox/utils/boxes.py:89-113."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[:, :2] - b[:, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[:, :2] + b[:, 2:] / 2)
    area_a = torch.prod(a[:, 2:], 1)
    area_b = torch.prod(b[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=2)
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)


def bboxes_iou_xyxy(a, b):
    """bboxes_iou(xyxy=True), boxes.py:94-98,111-113."""
    tl = torch.max(a[:, None, :2], b[:, :2])
    br = torch.min(a[:, None, 2:], b[:, 2:])
    area_a = torch.prod(a[:, 2:] - a[:, :2], 1)
    area_b = torch.prod(b[:, 2:] - b[:, :2], 1)
    en = (tl < br).type(tl.type()).prod(dim=2)
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)


def is_in_centers(gt, gx, gy, gs):
    """yolo_head.py:713-730: anchor centre within 1.5 strides of the GT centre.  [n, A] bool."""
    xc = ((gx + 0.5) * gs)[None]
    yc = ((gy + 0.5) * gs)[None]
    dist = (gs * 1.5)[None]
    l_ = gt[:, 0:1] - dist
    r_ = gt[:, 0:1] + dist
    t_ = gt[:, 1:2] - dist
    b_ = gt[:, 1:2] + dist
    deltas = torch.stack([xc - l_, yc - t_, r_ - xc, b_ - yc], 2)
    return deltas.min(dim=-1).values > 0.0


def simota_matching(cost, ious, num_gt):
    """yolo_head.py:734-774.  cost/ious: [n_gt, n_pos].  Returns (matching [n_gt,n_pos] uint8,
    fg_inboxes [n_pos] bool)."""
    matching = torch.zeros_like(cost, dtype=torch.uint8)
    k = min(10, ious.size(1))
    topk_ious, _ = torch.topk(ious, k, dim=1)
    dynamic_ks = torch.clamp(topk_ious.sum(1).int(), min=1)
    for g in range(num_gt):
        _, pos = torch.topk(cost[g], k=int(dynamic_ks[g]), largest=False)
        matching[g][pos] = 1
    amg = matching.sum(0)
    if amg.max() > 1:
        multi = amg > 1
        _, cmin = torch.min(cost[:, multi], dim=0)
        matching[:, multi] *= 0
        matching[cmin, multi] = 1
    return matching, amg > 0


@torch.no_grad()
def get_assignments(gt_boxes, gt_classes, valid_mask, boxes_pred, gx, gy, gs, cls_logits, obj_logits,
                    num_classes):
    """get_assignments / get_assignments_w_ignore (yolo_head.py:606-700, 974-1094) for one image.

    gt_boxes [n,4] cxcywh incl. ignore boxes; valid_mask [n] bool (False = ignore label).
    Returns dict(fg_mask[A] bool, ignore_mask[A] bool, matched_gt_inds[n_fg] (index among *valid*
    gts), gt_matched_classes[n_fg], pred_ious[n_fg], num_fg)."""
    inc = is_in_centers(gt_boxes, gx, gy, gs)                       # [n, A]
    anchor_filter = inc.sum(0) > 0
    if bool(valid_mask.all()):
        ignore_mask = torch.zeros_like(anchor_filter)
        geom = inc[:, anchor_filter]
    else:
        # yolo_head.py:1135-1146 (single pass == two-pass oracle commented at :1112-1116)
        anchor_filter_valid = inc[valid_mask].sum(0) > 0
        ignore_mask = anchor_filter & (~anchor_filter_valid)
        anchor_filter = anchor_filter.clone()
        anchor_filter[ignore_mask] = False
        geom = inc[valid_mask][:, anchor_filter]
    gt_b = gt_boxes[valid_mask]
    gt_c = gt_classes[valid_mask]
    n = gt_b.shape[0]
    fg_mask = anchor_filter.clone()
    bp = boxes_pred[fg_mask]
    ious = bboxes_iou_cxcywh(gt_b, bp)
    onehot = F.one_hot(gt_c.to(torch.int64), num_classes).float()
    iou_loss = -torch.log(ious + 1e-8)
    p = (cls_logits[fg_mask].float().sigmoid() * obj_logits[fg_mask].float().sigmoid()).sqrt()
    cls_loss = F.binary_cross_entropy(p.unsqueeze(0).repeat(n, 1, 1),
                                      onehot.unsqueeze(1).repeat(1, p.shape[0], 1),
                                      reduction='none').sum(-1)
    cost = cls_loss + 3.0 * iou_loss + float(1e6) * (~geom)
    matching, fg_in = simota_matching(cost, ious, n)
    fg_mask[fg_mask.clone()] = fg_in
    matched = matching[:, fg_in].argmax(0)
    return dict(fg_mask=fg_mask, ignore_mask=ignore_mask, matched_gt_inds=matched,
                gt_matched_classes=gt_c[matched], pred_ious=(matching * ious).sum(0)[fg_in],
                num_fg=int(fg_in.sum()), cost=cost, ious=ious)


def iou_loss_fn(pred, target):
    """IOUloss(reduction='none', loss_type='iou'), losses.py:18-43: 1 - iou^2 on cxcywh boxes."""
    tl = torch.max(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    br = torch.min(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    area_p = torch.prod(pred[:, 2:], 1)
    area_g = torch.prod(target[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=1)
    area_i = torch.prod(br - tl, 1) * en
    iou = area_i / (area_p + area_g - area_i + 1e-16)
    return 1 - iou ** 2


def sigmoid_focal_loss(inputs, targets, alpha=0.25, gamma=2.0):
    """torchvision.ops.sigmoid_focal_loss(reduction='none') as used by FocalLoss, losses.py:69-85."""
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction='none')
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    return (alpha * targets + (1 - alpha) * (1 - targets)) * loss


def ignore_bbox_(labels, ignore_bbox_thresh, ignore_label=1024):
    """_ignore_bbox, yolo_head.py:383-401 (mutates labels[:,:,0])."""
    if not ignore_bbox_thresh:
        return labels
    cls_idx = labels[:, :, 0]
    obj_conf, cls_conf = labels[:, :, 5], labels[:, :, 6]
    ign = torch.zeros_like(cls_idx).bool()
    for idx, th in enumerate(ignore_bbox_thresh):
        ign = ign | ((cls_idx == idx) & ((obj_conf < th) | (cls_conf < th)))
    ign = ign & (labels.sum(dim=2) > 0)
    labels[:, :, 0] = torch.where(ign, torch.full_like(cls_idx, ignore_label), cls_idx)
    return labels


def highest_score_mask(scores, k, exclude_mask=None):
    """_get_highest_score_mask, yolo_head.py:335-356: the top k (fraction) objectness logits of one image,
    counted and picked among the anchors outside ``exclude_mask`` (the foreground)."""
    if k <= 0:
        return None
    scores = scores.reshape(-1)
    if exclude_mask is not None and bool(exclude_mask.any()):
        n = int((~exclude_mask).float().sum().item() * k)
        ex = exclude_mask.reshape(-1).type_as(scores)
        scores = scores * (1. - ex) + ex * (-1e6)
    else:
        n = int(scores.shape[0] * k)
    mask = torch.zeros_like(scores).bool()
    if n == 0:
        return mask
    mask[scores.topk(n, dim=0, largest=True, sorted=False)[1]] = True
    return mask


def bbox_loss_weight(spec, matched_gt_inds, obj_conf, cls_conf):
    """_get_bbox_loss_weight, yolo_head.py:358-381: 'obj' | 'cls' | 'objxcls', optionally '-<expr of w>' ('cls-w**2')."""
    if not spec:
        return None
    val, expr = spec.split('-', 1) if '-' in spec else (spec, 'w')
    if val == 'obj':
        w = obj_conf[matched_gt_inds]
    elif val == 'cls':
        w = cls_conf[matched_gt_inds]
    elif val == 'objxcls':
        w = obj_conf[matched_gt_inds] * cls_conf[matched_gt_inds]
    else:
        raise NotImplementedError(spec)
    return eval(expr)  # noqa: S307  (the reference evaluates the configured expression the same way, :376)


def get_losses(gx, gy, gs, labels, outputs, num_classes=None, obj_focal_loss=False,
               reg_weight=5.0, obj_weight=1.0, cls_weight=1.0, ignore_bbox_thresh=None,
               ignore_label=1024, return_assign=False, bbox_loss_weighting='', ignore_bg_k=0):
    """get_losses / get_losses_w_ignore, yolo_head.py:403-597, 776-972 (use_l1 False).  ``bbox_loss_weighting``
    weighs the IoU and class terms of a foreground anchor by its box's confidence, normalised to batch mean 1
    (:523-524, :550-553, :903-904, :928-931); ``ignore_bg_k`` drops the top fraction of background objectness
    logits of each image from the objectness loss -- only on batches without ignore boxes (:541-542, :558-570;
    get_losses_w_ignore has no such step).

    labels [B,N,7] = (cls, cx, cy, w, h, obj_conf, cls_conf), zero rows = padding (assumed to be a
    suffix, :466); outputs [B,A,5+nc] = decoded boxes + obj/cls logits."""
    nc = outputs.shape[-1] - 5 if num_classes is None else num_classes
    labels = ignore_bbox_(labels, ignore_bbox_thresh, ignore_label)
    bbox_preds, obj_preds, cls_preds = outputs[:, :, :4], outputs[:, :, 4:5], outputs[:, :, 5:]
    nonzero = labels.sum(dim=2) > 0
    valid = labels[:, :, 0] != ignore_label
    nlabel = (nonzero & valid).sum(dim=1)
    nlabel_w = nonzero.sum(dim=1)
    A = outputs.shape[1]
    cls_t, reg_t, obj_t, fg_ms, ign_ms, assigns, bbox_ws = [], [], [], [], [], [], []
    num_fg, num_gts = 0.0, 0.0
    top_bg = ignore_bg_k > 0 and not bool((labels[:, :, 0] == ignore_label).any())
    for b in range(outputs.shape[0]):
        n = int(nlabel[b])
        num_gts += n
        if n == 0:
            cls_t.append(outputs.new_zeros((0, nc)))
            reg_t.append(outputs.new_zeros((0, 4)))
            obj_t.append(outputs.new_zeros((A, 1)))
            fg = outputs.new_zeros(A).bool()
            if labels[b].sum() == 0:
                ign = outputs.new_zeros(A).bool()
            else:  # only ignore boxes (:832-836): anchors inside them carry no obj loss
                n_ign = int((labels[b, :, 0] == ignore_label).sum())
                ign = is_in_centers(labels[b, :n_ign, 1:5], gx, gy, gs).sum(0) > 0
            fg_ms.append(fg)
            ign_ms.append(highest_score_mask(obj_preds[b].detach(), ignore_bg_k, fg) if top_bg else ign)
            assigns.append(None)
            continue
        nw = int(nlabel_w[b])
        res = get_assignments(labels[b, :nw, 1:5], labels[b, :nw, 0], valid[b, :nw],
                              bbox_preds[b].detach(), gx, gy, gs, cls_preds[b].detach(),
                              obj_preds[b].detach(), nc)
        num_fg += res['num_fg']
        cls_t.append(F.one_hot(res['gt_matched_classes'].to(torch.int64), nc) * res['pred_ious'].unsqueeze(-1))
        obj_t.append(res['fg_mask'].unsqueeze(-1).to(outputs.dtype))
        reg_t.append(labels[b, :nw, 1:5][valid[b, :nw]][res['matched_gt_inds']])
        fg_ms.append(res['fg_mask'])
        ign_ms.append(highest_score_mask(obj_preds[b].detach(), ignore_bg_k, res['fg_mask']) if top_bg else res['ignore_mask'])
        assigns.append(res)
        if bbox_loss_weighting:
            bbox_ws.append(bbox_loss_weight(bbox_loss_weighting, res['matched_gt_inds'], labels[b, :nw, 5][valid[b, :nw]],
                                            labels[b, :nw, 6][valid[b, :nw]]))
    cls_t, reg_t, obj_t = torch.cat(cls_t, 0), torch.cat(reg_t, 0), torch.cat(obj_t, 0)
    fg_ms, ign_ms = torch.cat(fg_ms, 0), torch.cat(ign_ms, 0)
    num_fg = max(num_fg, 1)
    fg_boxes = bbox_preds.reshape(-1, 4)[fg_ms]
    # IOUloss(reduction='mean'); returns the python float 0. when there is no fg (losses.py:20-21)
    w = 1.
    if bbox_loss_weighting and bbox_ws:
        w = torch.cat(bbox_ws, 0)
        w = w / w.mean()                                            # batch mean 1 (:552)
    loss_iou = (iou_loss_fn(fg_boxes, reg_t) * w).mean() if fg_boxes.shape[0] > 0 else outputs.new_zeros(())
    keep = ~ign_ms
    ol, ot = obj_preds.reshape(-1, 1)[keep], obj_t[keep]
    obj_l = sigmoid_focal_loss(ol, ot) if obj_focal_loss else \
        F.binary_cross_entropy_with_logits(ol, ot, reduction='none')
    loss_obj = obj_l.sum() / num_fg
    loss_cls = (F.binary_cross_entropy_with_logits(cls_preds.reshape(-1, nc)[fg_ms], cls_t, reduction='none') *
                (w[:, None] if torch.is_tensor(w) else w)).sum() / num_fg
    loss_iou = reg_weight * loss_iou
    loss_obj = obj_weight * loss_obj
    loss_cls = cls_weight * loss_cls
    out = dict(loss=loss_iou + loss_obj + loss_cls, iou_loss=loss_iou, conf_loss=loss_obj,
               cls_loss=loss_cls, l1_loss=0.0, num_fg=num_fg / max(num_gts, 1))
    if return_assign:
        out['_assign'] = assigns
        out['_fg_mask'] = fg_ms.view(outputs.shape[0], A)
        out['_ignore_mask'] = ign_ms.view(outputs.shape[0], A)
    return out


def detect_forward(feats, sd, cfg, labels=None, training=False):
    """YoloXDetector.forward_detect, models/detection/yolox_extension/models/detector.py:55-77.
    cfg: dict(n_bottleneck, strides, in_stages, + loss kwargs)."""
    fpn = pafpn_forward(feats, sd, cfg['n_bottleneck'], tuple(cfg.get('in_stages', (2, 3, 4))),
                        training=training)
    kw = {k: cfg[k] for k in ('obj_focal_loss', 'ignore_bbox_thresh', 'ignore_label', 'bbox_loss_weighting', 'ignore_bg_k') if k in cfg}
    return head_forward(fpn, sd, cfg['strides'], labels=labels, training=training, **kw)


def init_prior_bias(prior_prob=0.01):
    """initialize_biases, yolo_head.py:184-193."""
    return -math.log((1 - prior_prob) / prior_prob)
