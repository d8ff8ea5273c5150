"""fetch_model_module / fetch_data_module with the reference's dispatch (modules/utils/fetch.py:10-38)."""


def fetch_model_module(config):
    name = config.model.name
    if name == 'rnndet':
        if config.get('tta', {}).get('enable', False):
            from leod_amd.modules.utils.tta import TTAModule
            return TTAModule(config)
        from leod_amd.modules.detection import Module
        return Module(config)
    if name == 'pseudo_labeler':
        from leod_amd.modules.pseudo_labeler import PseudoLabeler
        return PseudoLabeler(config)
    raise NotImplementedError(name)


def fetch_data_module(config, **loader_kw):
    """The data module of ``config.dataset`` with the batch sizes / worker counts of ``config.batch_size`` / ``config.hardware.num_workers``
    (fetch.py:22-38; per rank, as under Lightning DDP).  ``loader_kw``: the loader options this package adds (``worker_process``, ``device``,
    ``prefetch``, ``pin_memory``, ``rank`` / ``world_size``)."""
    generic = config.hardware.get('num_workers', None)
    nw = config.hardware.num_workers
    nw_train = nw.get('train', generic) if hasattr(nw, 'get') else int(nw)
    nw_eval = nw.get('eval', generic) if hasattr(nw, 'get') else int(nw)
    if config.dataset.name in ('gen1', 'gen4'):
        from leod_amd.modules.data.genx import DataModule
        return DataModule(config.dataset, num_workers_train=nw_train, num_workers_eval=nw_eval, batch_size_train=config.batch_size.train,
                          batch_size_eval=config.batch_size.eval, **loader_kw)
    raise NotImplementedError(config.dataset.name)
