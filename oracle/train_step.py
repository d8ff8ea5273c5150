"""Oracle: one full training step / one inference pass of the LEOD detector on CPU (fp32,
torch autograd).  TEST INFRASTRUCTURE and the ``cpu_baseline`` ("port") leg of bench.py.

Follows modules/detection.py:150-298 (training_step), :485-518 (AdamW + OneCycleLR),
train.py:236-243 (gradient clip *by value* 1.0) and modules/pseudo_labeler.py:622-770 for the
inference loop.  Citations relative to /root/reference.
"""
from typing import Dict, List, Optional

import torch

from . import backbone as ob
from . import head as oh
from . import postproc as op


def model_cfg(embed_dim=48, dim_head=24, fpn_depth=0.33, partition_size=(8, 10), num_classes=2,
              in_res_hw=(256, 320)):
    dims = [embed_dim * m for m in (1, 2, 4, 8)]
    return dict(embed_dim=embed_dim, dim_head=dim_head, partition_size=tuple(partition_size),
                num_blocks=(1, 1, 1, 1), patch_size=4, dims=dims, n_bottleneck=round(3 * fpn_depth),
                strides=(8, 16, 32), in_stages=(2, 3, 4), num_classes=num_classes,
                in_res_hw=tuple(in_res_hw))


def forward_sequence(sd, cfg, ev_seq, labels, prev_states, is_first_sample=None, training=True):
    """ev_seq [T,B,C,H,W] (uint8/float, unpadded); labels[t][b] = None | [n,8] tensor.
    Returns (losses|None, decoded predictions, final states, selected (t,b) list)."""
    T, B = ev_seq.shape[:2]
    ev = op_pad(ev_seq.to(torch.float32), cfg['in_res_hw'])               # detection.py:132-135
    if prev_states is not None and is_first_sample is not None and bool(is_first_sample.any()):
        # RNNStates.reset, modules/utils/detection.py:120-139,152-157
        prev_states = [(h.detach().clone(), c.detach().clone()) for h, c in prev_states]
        for h, c in prev_states:
            h[is_first_sample] = 0
            c[is_first_sample] = 0
    sel_feats: Dict[int, List[torch.Tensor]] = {}
    sel_labels, sel_idx = [], []
    states = prev_states
    for t in range(T):
        feats, states = ob.backbone_forward(ev[t], states, sd, cfg)
        idx = [b for b in range(B) if labels[t][b] is not None and len(labels[t][b]) > 0]
        if idx:                                                            # detection.py:209-224
            for k, v in feats.items():
                sel_feats.setdefault(k, []).append(v[idx])
            sel_labels.extend(labels[t][b] for b in idx)
            sel_idx.extend((t, b) for b in idx)
    feats = {k: torch.cat(v, 0) for k, v in sel_feats.items()}
    targets = op.batched_yolox_labels(sel_labels) if training else None
    preds, losses = oh.detect_forward(feats, sd, cfg, labels=targets, training=training)
    return losses, preds, states, sel_idx


def op_pad(ev, hw):
    return ob.pad_ev_repr(ev, hw)


class OracleTrainer:
    """Holds leaf parameters + torch.optim.AdamW + OneCycleLR exactly as configure_optimizers does
    (detection.py:485-518) and clips gradients by value (train.py:236-237)."""

    def __init__(self, state_dict, cfg, lr=2e-4, weight_decay=0.0, total_steps=400000, pct_start=0.005,
                 div_factor=20, final_div_factor=10000, clip_value=1.0):
        self.cfg = cfg
        self.sd = {k: v.clone() for k, v in state_dict.items()}
        self.param_keys = [k for k, v in self.sd.items()
                           if v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))]
        for k in self.param_keys:
            self.sd[k].requires_grad_(True)
        self.opt = torch.optim.AdamW([self.sd[k] for k in self.param_keys], lr=lr, weight_decay=weight_decay)
        self.sched = torch.optim.lr_scheduler.OneCycleLR(
            self.opt, max_lr=lr, div_factor=div_factor, final_div_factor=final_div_factor / div_factor,
            total_steps=total_steps, pct_start=pct_start, cycle_momentum=False, anneal_strategy='linear')
        self.clip_value = clip_value
        self.states = None

    def step(self, ev_seq, labels, is_first_sample=None):
        self.opt.zero_grad(set_to_none=True)
        losses, preds, states, _ = forward_sequence(self.sd, self.cfg, ev_seq, labels, self.states,
                                                    is_first_sample, training=True)
        losses['loss'].backward()
        if self.clip_value:
            torch.nn.utils.clip_grad_value_([self.sd[k] for k in self.param_keys], self.clip_value)
        grads = {k: (self.sd[k].grad.clone() if self.sd[k].grad is not None else None) for k in self.param_keys}
        self.opt.step()
        self.sched.step()
        self.states = [(h.detach(), c.detach()) for h, c in states]       # save_states_and_detach
        return {k: (float(v.detach()) if torch.is_tensor(v) else v) for k, v in losses.items()}, grads


@torch.no_grad()
def infer_sequence(sd, cfg, ev_seq, prev_states=None, conf_thre=0.01, nms_thre=0.45, hflip=False,
                   device_semantics='gpu'):
    """Pseudo-label style inference (pseudo_labeler.py:458-495,676-704,565-589): optional hflip copy
    concatenated on the batch dim, backbone over T, head + postprocess on every frame.
    Returns list over (t, b') of [n,7] detections and the final states."""
    ev = ev_seq.to(torch.float32)
    if hflip:
        ev = torch.cat([ev, torch.flip(ev, dims=[-1])], dim=1)
    ev = op_pad(ev, cfg['in_res_hw'])
    T = ev.shape[0]
    states = prev_states
    feats_all: Dict[int, List[torch.Tensor]] = {}
    for t in range(T):
        feats, states = ob.backbone_forward(ev[t], states, sd, cfg)
        for k, v in feats.items():
            feats_all.setdefault(k, []).append(v)
    feats = {k: torch.cat(v, 0) for k, v in feats_all.items()}
    preds, _ = oh.detect_forward(feats, sd, cfg, training=False)
    dets = op.postprocess(preds, cfg['num_classes'], conf_thre, nms_thre,
                          pad=torch.zeros((0, 7)), device_semantics=device_semantics)
    return dets, states, preds
