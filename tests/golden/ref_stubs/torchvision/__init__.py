from . import ops, transforms  # noqa
