"""Config container: omegaconf's DictConfig when it is installed (the reference's Hydra surface),
otherwise a small attribute-dict with the handful of calls the hot path makes (.get, attribute
access, to_container).  Keys are the reference's (config/model/maxvit_yolox/default.yaml etc.)."""
from typing import Any, Mapping

try:  # pragma: no cover - not installed in the build image
    from omegaconf import DictConfig, OmegaConf, open_dict  # type: ignore

    def to_container(cfg) -> dict:
        return OmegaConf.to_container(cfg, resolve=True, throw_on_missing=True)

    def create(d: Mapping) -> Any:
        return OmegaConf.create(dict(d))

    HAVE_OMEGACONF = True
except ImportError:
    HAVE_OMEGACONF = False

    class DictConfig(dict):
        def __init__(self, d: Mapping = None):
            super().__init__()
            for k, v in (d or {}).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, Mapping) and not isinstance(v, DictConfig):
                v = DictConfig(v)
            super().__setitem__(k, v)

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    class open_dict:  # noqa: N801
        def __init__(self, cfg):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def to_container(cfg) -> dict:
        return {k: (to_container(v) if isinstance(v, Mapping) else (list(v) if isinstance(v, tuple) else v))
                for k, v in cfg.items()}

    def create(d: Mapping) -> Any:
        return DictConfig(d)
