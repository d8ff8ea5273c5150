from .dictconfig import DictConfig, create, to_container, open_dict  # noqa: F401
from .defaults import model_config, full_config  # noqa: F401
from .modifier import dynamically_modify_train_config  # noqa: F401
