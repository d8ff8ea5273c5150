"""fetch_model_module with the reference's dispatch (modules/utils/fetch.py:10-19)."""


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
