"""Mirror of the helpers of the reference's utils/helpers.py that the hot path uses."""
import torch as th


def torch_uniform_sample_scalar(min_value: float, max_value: float):
    """utils/helpers.py:13-17: one ``th.rand(1)`` draw unless the interval is degenerate."""
    assert max_value >= min_value, f'{max_value=} is smaller than {min_value=}'
    if max_value == min_value:
        return min_value
    return min_value + (max_value - min_value) * th.rand(1).item()
