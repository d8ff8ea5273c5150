from .maxvit_rnn import RNNDetector as MaxViTRNNDetector


def build_recurrent_backbone(backbone_cfg):
    if backbone_cfg.name == 'MaxViTRNN':
        return MaxViTRNNDetector(backbone_cfg)
    raise NotImplementedError(backbone_cfg.name)
