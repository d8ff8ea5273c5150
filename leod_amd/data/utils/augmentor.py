"""Spatial data augmentation of the GenX loaders, on the device (reference: data/utils/augmentor.py:27-558).

What is random stays on the host and draws from torch's RNG in the reference's order, so a seed gives the same
``AugmentationState`` as the reference's ``RandomSpatialAugmentorGenX`` (tests/golden/g14_augment.npz).  What touches
pixels -- flip, zoom-in (crop + nearest-exact resize), zoom-out (nearest-exact resize + paste) -- is ONE gather kernel over
the uint8 representation of a whole batch of sequences (``leod_augment_u8``): the reference runs ``flip`` /
``interpolate`` / slice assignment per sample and timestep in the dataloader workers.  Labels are transformed by the
``ObjectLabels`` methods mirrored from data/genx_utils/labels.py.

Rotation (probability 0 in every shipped config) and the flow / image data types are not part of the LEOD path.
"""
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch as th

from leod_amd._lib import lib, check
from leod_amd.data.genx_utils.labels import ObjectLabels
from leod_amd.utils.helpers import torch_uniform_sample_scalar


@dataclass
class ZoomOutState:
    active: bool = False
    x0: int = 0
    y0: int = 0
    zoom_out_factor: float = 1.0


@dataclass
class ZoomInState:
    active: bool = False
    x0: int = 0
    y0: int = 0
    zoom_in_factor: float = 1.0


@dataclass
class RotationState:
    active: bool = False
    angle_deg: float = 0.0


@dataclass
class AugmentationState:
    apply_h_flip: bool = False
    apply_t_flip: bool = False
    rotation: RotationState = field(default_factory=RotationState)
    zoom_in: ZoomInState = field(default_factory=ZoomInState)
    zoom_out: ZoomOutState = field(default_factory=ZoomOutState)

    def to_dict(self) -> Dict[str, Any]:
        return dict(h_flip=dict(active=self.apply_h_flip), t_flip=dict(active=self.apply_t_flip),
                    rotation=vars(self.rotation).copy(), zoom_in=vars(self.zoom_in).copy(),
                    zoom_out=vars(self.zoom_out).copy())


def randomly_sample_zoom_window_from_label_rectangle(label_xywh, input_height, input_width, zoom_window_height,
                                                     zoom_window_width) -> Tuple[int, int]:
    """Top-left corner of a zoom window that contains the whole label (augmentor.py:521-558): two uniform draws."""
    assert input_height >= zoom_window_height and input_width >= zoom_window_width
    x0_l, y0_l, w_l, h_l = tuple(v.item() if isinstance(v, th.Tensor) else v for v in label_xywh)
    x1_l, y1_l = x0_l + w_l, y0_l + h_l
    assert x0_l >= 0 and y0_l >= 0 and w_l > 0 and h_l > 0
    x0_valid = max(x1_l - max(zoom_window_width, w_l), 0)
    y0_valid = max(y1_l - max(zoom_window_height, h_l), 0)
    x1_valid = min(x0_l + max(zoom_window_width, w_l), input_width - 1)
    y1_valid = min(y0_l + max(zoom_window_height, h_l), input_height - 1)
    x1_valid = max(x1_valid - zoom_window_width, x0_valid)
    y1_valid = max(y1_valid - zoom_window_height, y0_valid)
    x = int(torch_uniform_sample_scalar(min_value=x0_valid, max_value=x1_valid))
    y = int(torch_uniform_sample_scalar(min_value=y0_valid, max_value=y1_valid))
    assert 0 <= x < input_width and 0 <= y < input_height
    return x, y


def randomly_sample_zoom_window_from_objframe(objframe: ObjectLabels, zoom_window_height, zoom_window_width) -> Tuple[int, int]:
    """augmentor.py:495-518 (including its ``randint(high=len-1)`` choice, which never picks the last label)."""
    input_height, input_width = objframe.input_size_hw
    samples = [randomly_sample_zoom_window_from_label_rectangle(
        (objframe.x[i], objframe.y[i], objframe.w[i], objframe.h[i]), input_height, input_width, zoom_window_height,
        zoom_window_width) for i in range(len(objframe))]
    assert len(samples) > 0
    idx = 0 if len(samples) == 1 else th.randint(low=0, high=len(samples) - 1, size=(1,)).item()
    return samples[idx]


def get_most_recent_objframe(labels: Sequence[Optional[ObjectLabels]], check_if_nonempty: bool = True) -> Optional[ObjectLabels]:
    for lab in reversed(labels):
        if lab is not None and (not check_if_nonempty or len(lab) > 0):
            return lab
    return None


class RandomSpatialAugmentorGenX:
    """Same constructor, config keys and random draws as the reference class (augmentor.py:125-207, 284-309); the pixel
    work of a whole batch is applied afterwards by ``augment_batch``."""

    def __init__(self, dataset_hw: Tuple[int, int], automatic_randomization: bool, augm_config):
        assert isinstance(dataset_hw, tuple) and len(dataset_hw) == 2 and all(x > 0 for x in dataset_hw)
        self.hw_tuple = dataset_hw
        self.automatic_randomization = automatic_randomization
        self.h_flip_prob = augm_config.prob_hflip
        self.t_flip_prob = augm_config.prob_tflip
        self.rot_prob = augm_config.rotate.prob
        self.rot_min_angle_deg = augm_config.rotate.get('min_angle_deg', 0)
        self.rot_max_angle_deg = augm_config.rotate.max_angle_deg
        self.zoom_prob = augm_config.zoom.prob
        zoom_out_weight = augm_config.zoom.zoom_out.get('weight', 1)
        self.min_zoom_out_factor = augm_config.zoom.zoom_out.factor.min
        self.max_zoom_out_factor = augm_config.zoom.zoom_out.factor.max
        has_zoom_in = 'zoom_in' in augm_config.zoom
        zoom_in_weight = augm_config.zoom.zoom_in.weight if has_zoom_in else 0
        self.min_zoom_in_factor = augm_config.zoom.zoom_in.factor.min if has_zoom_in else 1
        self.max_zoom_in_factor = augm_config.zoom.zoom_in.factor.max if has_zoom_in else 1
        assert 0 <= self.h_flip_prob <= 1 and 0 <= self.t_flip_prob <= 1 and 0 <= self.rot_prob <= 1
        assert 0 <= self.zoom_prob <= 1 and zoom_in_weight >= 0 and zoom_out_weight >= 0
        assert self.max_zoom_in_factor >= self.min_zoom_in_factor >= 1
        assert self.max_zoom_out_factor >= self.min_zoom_out_factor >= 1
        self.zoom_in_or_out_distribution = torch.distributions.categorical.Categorical(
            probs=th.tensor([zoom_in_weight, zoom_out_weight]))
        self.augm_state = AugmentationState()

    def randomize_augmentation(self) -> None:
        """Label-independent part of the state (augmentor.py:173-207), same RNG call sequence."""
        st = self.augm_state
        st.apply_h_flip = self.h_flip_prob > th.rand(1).item()
        st.apply_t_flip = self.t_flip_prob > th.rand(1).item()
        st.rotation.active = self.rot_prob > th.rand(1).item()
        if st.rotation.active:
            sign = 1 if th.randn(1).item() >= 0 else -1
            st.rotation.angle_deg = sign * torch_uniform_sample_scalar(self.rot_min_angle_deg, self.rot_max_angle_deg)
        do_zoom = self.zoom_prob > th.rand(1).item()
        do_zoom_in = self.zoom_in_or_out_distribution.sample().item() == 0
        do_zoom_out = not do_zoom_in
        do_zoom_in &= do_zoom
        do_zoom_out &= do_zoom
        st.zoom_in.active, st.zoom_out.active = do_zoom_in, do_zoom_out
        if do_zoom_out:
            f = torch_uniform_sample_scalar(self.min_zoom_out_factor, self.max_zoom_out_factor)
            height, width = self.hw_tuple
            win_h, win_w = int(height / f), int(width / f)
            st.zoom_out.x0 = int(torch_uniform_sample_scalar(0, width - win_w))
            st.zoom_out.y0 = int(torch_uniform_sample_scalar(0, height - win_h))
            st.zoom_out.zoom_out_factor = f

    def sample_zoom_in(self, labels: Sequence[Optional[ObjectLabels]]) -> None:
        """Label-dependent part (augmentor.py:284-309): the zoom window must contain a label of the most recent
        labelled frame; without labels the zoom-in is switched off."""
        st = self.augm_state.zoom_in
        f = torch_uniform_sample_scalar(self.min_zoom_in_factor, self.max_zoom_in_factor)
        if f == 1:
            st.active, st.x0, st.y0, st.zoom_in_factor = False, 0, 0, 1
            return
        height, width = self.hw_tuple
        win_h, win_w = int(height / f), int(width / f)
        latest = get_most_recent_objframe(labels, check_if_nonempty=True)
        if latest is None:
            st.active, st.x0, st.y0, st.zoom_in_factor = False, 0, 0, 1
            return
        st.x0, st.y0 = randomly_sample_zoom_window_from_objframe(latest, win_h, win_w)
        st.zoom_in_factor = f


    def __call__(self, data_dict: Dict[Any, Any]) -> Dict[Any, Any]:
        """One loader sample (reference ``__call__``, augmentor.py:455-476): the labels (visible and withheld) are transformed
        here on the host; the event frames are NOT touched -- the drawn state rides along under ``DataType.AUGM_STATE`` and
        the whole batch is flipped / zoomed by one ``leod_augment_u8`` launch on the device (``augment_events``)."""
        from leod_amd.data.utils.types import DataType
        primary = data_dict[DataType.OBJLABELS_SEQ]
        others = [data_dict[k] for k in (DataType.SKIPPED_OBJLABELS_SEQ,) if k in data_dict]
        # the two containers of the padding sample are one object: transform it once
        others = [o for o in others if o is not primary]
        state = self.augment_sample_labels(primary.sparse_object_labels_batch,
                                           extra=[o.sparse_object_labels_batch for o in others])
        if state.zoom_in.active:                     # boxes outside the zoom window are gone: empty frames become None
            for c in [primary] + others:
                c.sparse_object_labels_batch[:] = [None if (l is not None and len(l) == 0) else l for l in c.sparse_object_labels_batch]
        data_dict[DataType.AUGM_STATE] = state
        return data_dict

    def augment_sample_labels(self, labels: Sequence[Optional[ObjectLabels]],
                              extra: Sequence[Sequence[Optional[ObjectLabels]]] = ()) -> AugmentationState:
        """Everything ``__call__`` of the reference does for ONE loader sample except the pixel work (augmentor.py:455-476,
        in that order): draw the state, flip the labels, sample the zoom-in window from the (flipped) labels, transform
        the labels (``extra``: further label lists of the sample that follow the same transform, e.g. the withheld labels).
        Returns a copy of the resulting state; feed the states of a batch to ``augment_events``."""
        import copy
        every = [labels] + [list_ for list_ in extra]
        if self.automatic_randomization:
            self.randomize_augmentation()
        st = self.augm_state
        assert not st.apply_t_flip, 'should do this outside this class (due to streaming loading mode)'
        if st.rotation.active:
            raise NotImplementedError('rotation augmentation (probability 0 in every shipped config)')
        if st.apply_h_flip:
            for lab in (l for list_ in every for l in list_):
                if lab is not None:
                    lab.flip_lr_()
        if st.zoom_in.active:
            self.sample_zoom_in(labels)
            if st.zoom_in.active:
                for lab in (l for list_ in every for l in list_):
                    if lab is not None:
                        lab.zoom_in_and_rescale_((st.zoom_in.x0, st.zoom_in.y0), st.zoom_in.zoom_in_factor)
        if st.zoom_out.active:
            assert not st.zoom_in.active
            if st.zoom_out.zoom_out_factor == 1:
                st.zoom_out.active, st.zoom_out.x0, st.zoom_out.y0 = False, 0, 0
            else:
                for lab in (l for list_ in every for l in list_):
                    if lab is not None:
                        lab.zoom_out_and_rescale_((st.zoom_out.x0, st.zoom_out.y0), st.zoom_out.zoom_out_factor)
        return copy.deepcopy(st)


def state_to_params(state: AugmentationState, hw: Tuple[int, int]) -> List[int]:
    """{hflip, mode, x0, y0, win_h, win_w, tflip} of ``leod_augment_u8``; window sizes as the reference computes them
    (``int(size / factor)``, cropped at the frame border like the slice ``[y0:y0+h, x0:x0+w]``).  ``apply_t_flip`` is
    the loader's ``time_flip_data`` (sequence_base.py:207-227: frames reversed, channel planes reversed) -- the reference's
    augmentor leaves it to the dataset (augmentor.py:464-465), here it rides in the same gather pass."""
    if state.rotation.active:
        raise NotImplementedError('rotation augmentation (probability 0 in every shipped config)')
    return _spatial_params(state, hw) + [int(bool(state.apply_t_flip))]


def _spatial_params(state: AugmentationState, hw: Tuple[int, int]) -> List[int]:
    H, W = hw
    if state.zoom_in.active and state.zoom_in.zoom_in_factor != 1:
        assert not state.zoom_out.active
        f = state.zoom_in.zoom_in_factor
        wh, ww = int(H / f), int(W / f)
        x0, y0 = state.zoom_in.x0, state.zoom_in.y0
        assert 0 <= x0 < W and 0 <= y0 < H and wh > 0 and ww > 0, (x0, y0, wh, ww)
        return [int(state.apply_h_flip), 1, x0, y0, min(wh, H - y0), min(ww, W - x0)]
    if state.zoom_out.active and state.zoom_out.zoom_out_factor != 1:
        f = state.zoom_out.zoom_out_factor
        wh, ww = int(H / f), int(W / f)
        x0, y0 = state.zoom_out.x0, state.zoom_out.y0
        assert 0 <= x0 and x0 + ww <= W and 0 <= y0 and y0 + wh <= H and wh > 0 and ww > 0, (x0, y0, wh, ww)
        return [int(state.apply_h_flip), 2, x0, y0, wh, ww]
    return [int(state.apply_h_flip), 0, 0, 0, H, W]


def augment_events(ev_seq: th.Tensor, states: Sequence[AugmentationState]) -> th.Tensor:
    """ev_seq [T,B,C,H,W] uint8 on the device, one state per batch sample -> augmented copy (one kernel launch)."""
    assert ev_seq.dim() == 5 and ev_seq.dtype == th.uint8 and ev_seq.is_cuda and ev_seq.is_contiguous()
    T, B, C, H, W = ev_seq.shape
    assert len(states) == B
    params = th.tensor([state_to_params(s, (H, W)) for s in states], dtype=th.int32).to(ev_seq.device, non_blocking=True)
    out = th.empty_like(ev_seq)
    check(lib().leod_augment_u8(ev_seq.data_ptr(), out.data_ptr(), params.data_ptr(), T, B, C, H, W,
                                th.cuda.current_stream().cuda_stream), 'augment_u8')
    return out


def augment_labels(labels: Sequence[Optional[ObjectLabels]], state: AugmentationState) -> List[Optional[ObjectLabels]]:
    """The label side of ``RandomSpatialAugmentorGenX.__call__`` for the frames of ONE sample (in place, like the
    reference): flip, then zoom-in or zoom-out (augmentor.py:455-476)."""
    for lab in labels:
        if lab is None:
            continue
        if state.apply_h_flip:
            lab.flip_lr_()
        if state.zoom_in.active:
            lab.zoom_in_and_rescale_((state.zoom_in.x0, state.zoom_in.y0), state.zoom_in.zoom_in_factor)
        if state.zoom_out.active:
            lab.zoom_out_and_rescale_((state.zoom_out.x0, state.zoom_out.y0), state.zoom_out.zoom_out_factor)
    return list(labels)
