"""Streaming datasets over a split directory (the interface of the reference's data/genx_utils/dataset_streaming.py:23-228):
training = sub-sequences with guaranteed labels, concatenated in random order per batch slot; validation / test /
pseudo-labelling = whole recordings dealt to ranks, workers and batch slots."""
import copy
import os
import pickle
from collections import defaultdict
from pathlib import Path
from typing import List, Optional, Union

from leod_amd.data.genx_utils.sequence_streaming import SequenceForIter
from leod_amd.data.utils.augmentor import RandomSpatialAugmentorGenX
from leod_amd.data.utils.stream_concat_datapipe import ConcatStreamingDataPipe
from leod_amd.data.utils.stream_sharded_datapipe import ShardedStreamingDataPipe
from leod_amd.data.utils.types import DatasetMode, DatasetType

MODE_2_STR = {DatasetMode.TRAIN: 'train', DatasetMode.VALIDATION: 'val', DatasetMode.TESTING: 'test'}
SPLITS_DIR = os.path.join(os.path.dirname(os.path.realpath(__file__)), 'splits')


def resolve_split_path(dataset_config, dataset_mode: DatasetMode) -> Path:
    dataset_path = Path(dataset_config.path)
    assert dataset_path.is_dir(), f'{dataset_path}'
    split_path = dataset_path / MODE_2_STR[dataset_mode]
    while split_path.is_symlink():
        split_path = Path(os.readlink(str(split_path)))
    assert split_path.is_dir(), f'{split_path}'
    return split_path


def subsample_sequence(split_path: Path, ratio: float) -> List[Path]:
    """Every k-th recording of the (sorted) split so that round(n * ratio) remain (utils/preprocessing.py:18-28)."""
    seq_dirs = sorted(p for p in Path(split_path).iterdir())
    if 0. < ratio < 1.:
        num = round(len(seq_dirs) * ratio)
        assert 1 <= num <= len(seq_dirs)
        seq_dirs = seq_dirs[0::len(seq_dirs) // num][:num]
        assert len(seq_dirs) == num
    return seq_dirs


def label_list_fn(dataset_config) -> str:
    """The per-recording list of visible labelled frames of the sparse-label (WSOD) regimes: ``splits/<name>/ssod_<ratio>-off0.pkl``,
    a pickled ``{recording name: [objframe idx]}`` (written by ``dataset_rnd.build_random_access_dataset``)."""
    return os.path.join(SPLITS_DIR, dataset_config.name, f'ssod_{dataset_config.ratio:.3f}-off0.pkl')


def dataset_type_of(dataset_config) -> DatasetType:
    return {'gen1': DatasetType.GEN1, 'gen4': DatasetType.GEN4}[dataset_config.name]


def get_sequences(path: Path, dataset_config, guarantee_labels: bool, label_list: Optional[List[int]] = None,
                  sparse_ratio: float = -1.0) -> List[SequenceForIter]:
    kw = dict(path=path, ev_representation_name=dataset_config.ev_repr_name, sequence_length=dataset_config.sequence_length,
              dataset_type=dataset_type_of(dataset_config), downsample_by_factor_2=dataset_config.downsample_by_factor_2,
              tflip_offset=dataset_config.data_augmentation.tflip_offset)
    if guarantee_labels:
        assert sparse_ratio == -1., 'cannot guarantee label when loading sparse labels in stream mode'
        return SequenceForIter.get_sequences_with_guaranteed_labels(**kw)
    return [SequenceForIter(objframe_idx=label_list, data_ratio=sparse_ratio,
                            start_from_zero=dataset_config.data_augmentation.stream.start_from_zero, **kw)]


def stream_augmentor_factory(dataset_config):
    hw = tuple(dataset_config.resolution_hw)
    if dataset_config.downsample_by_factor_2:
        hw = tuple(x // 2 for x in hw)
    if dataset_config.get('ssod', False):
        raise NotImplementedError('SSODAugmentorGenX (teacher-student SSOD training) is not part of the LEOD self-training path')
    cfg = dataset_config.data_augmentation.stream
    return lambda: RandomSpatialAugmentorGenX(dataset_hw=hw, automatic_randomization=False, augm_config=cfg)


def build_streaming_dataset(dataset_mode: DatasetMode, dataset_config, batch_size: int, num_workers: int,
                            pseudo_labeling: bool = False) -> Union[ConcatStreamingDataPipe, ShardedStreamingDataPipe]:
    split_path = resolve_split_path(dataset_config, dataset_mode)
    guarantee_labels = dataset_mode == DatasetMode.TRAIN
    sparse_ratio, label_lists = -1., defaultdict(lambda: None)
    if dataset_mode == DatasetMode.TRAIN:
        if pseudo_labeling:
            guarantee_labels = False                            # every frame is to be labelled
        if 0. < dataset_config.ratio < 1.:                      # sparse labels (WSOD): whole recordings, listed labels only
            guarantee_labels = False
            sparse_ratio = dataset_config.ratio
            with open(label_list_fn(dataset_config), 'rb') as f:
                label_lists = defaultdict(lambda: None, pickle.load(f))
        seq_dirs = subsample_sequence(split_path, dataset_config.train_ratio)
        if pseudo_labeling and 0. < dataset_config.train_ratio < 1.:
            # SSOD: the recordings left out of training are pseudo-labelled too, with ALL their labels withheld
            everything = subsample_sequence(split_path, -1)
            for entry in everything:
                if entry not in seq_dirs:
                    label_lists[entry.name] = []
            seq_dirs = everything
    elif dataset_mode == DatasetMode.VALIDATION:
        seq_dirs = subsample_sequence(split_path, dataset_config.val_ratio)
    elif dataset_mode == DatasetMode.TESTING:
        seq_dirs = subsample_sequence(split_path, dataset_config.test_ratio)
    else:
        raise NotImplementedError(f'Unknown dataset mode: {dataset_mode}')
    datapipes: List[SequenceForIter] = []
    for entry in seq_dirs:
        new = [dp for dp in get_sequences(entry, dataset_config, guarantee_labels, label_lists[entry.name], sparse_ratio) if len(dp) > 0]
        datapipes.extend(new)
    tta = dataset_config.get('tta', None)
    if tta is not None and tta.enable and tta.tflip:            # every recording a second time, reversed
        assert not dataset_config.reverse_event_order
        flipped = copy.deepcopy(datapipes)
        for dp in flipped:
            dp.time_flip = True
        datapipes.extend(flipped)
    if dataset_config.reverse_event_order:
        for dp in datapipes:
            dp.time_flip = True
    if dataset_config.only_load_labels:
        for dp in datapipes:
            dp.only_load_labels()
    assert len(datapipes) > 0
    if dataset_mode == DatasetMode.TRAIN and not pseudo_labeling:
        return ConcatStreamingDataPipe(datapipe_list=datapipes, batch_size=batch_size, num_workers=num_workers,
                                       augmentation_pipeline=stream_augmentor_factory(dataset_config))
    return ShardedStreamingDataPipe(datapipe_list=datapipes, batch_size=batch_size,
                                    fill_value=datapipes[0].get_fully_padded_sample())
