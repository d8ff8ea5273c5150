"""Mirror of the helpers of the reference's utils/helpers.py that the hot path uses."""
import torch as th


def torch_uniform_sample_scalar(min_value: float, max_value: float):
    """utils/helpers.py:13-17: one ``th.rand(1)`` draw unless the interval is degenerate."""
    assert max_value >= min_value, f'{max_value=} is smaller than {min_value=}'
    if max_value == min_value:
        return min_value
    return min_value + (max_value - min_value) * th.rand(1).item()


def th_cat(tensor_lst, dim: int = 0):
    """``th.cat`` that accepts an empty list (utils/helpers.py:7-10)."""
    return th.cat(tensor_lst, dim=dim) if len(tensor_lst) else th.tensor([])


def clamp(value, smallest, largest):
    return max(smallest, min(value, largest))


def subsample_list(lst, num: int, offset: int = 0):
    """``num`` items of ``lst``, every (len // num)-th one from ``offset`` (utils/helpers.py:24-29)."""
    assert len(lst) >= num >= 1, f'{len(lst)=} {num=}'
    return lst[offset::len(lst) // num][:num]


def list2d_to_list1d(lst_2d):
    """[L][B] -> ([L * B], row lengths); anything else is handed back with None (utils/helpers.py:32-40)."""
    if not (isinstance(lst_2d, list) and len(lst_2d) and isinstance(lst_2d[0], list)):
        return lst_2d, None
    return [item for row in lst_2d for item in row], [len(row) for row in lst_2d]


def list1d_to_list2d(lst_1d, lst_lens=None):
    if lst_lens is None or not isinstance(lst_lens, list):
        return lst_1d
    out, start = [], 0
    for n in lst_lens:
        out.append(lst_1d[start:start + n])
        start += n
    return out


def temporal_wrapper(func):
    """Lets a function written for flat lists take [L][B] lists (utils/helpers.py:55-104): every list-of-lists argument is flattened, the
    function runs once, and every LIST it returns is cut back into the rows of the corresponding flattened argument (in order)."""
    import inspect

    def wrapped(*args, **kwargs):
        bound = inspect.signature(func).bind(*args, **kwargs)
        bound.apply_defaults()
        flat, lens = [], []
        for value in bound.arguments.values():
            v1, ln = list2d_to_list1d(value)
            flat.append(v1)
            if ln is not None:
                lens.append(ln)
        out = func(*flat)
        outs = out if isinstance(out, tuple) else (out,)
        res, used = [], 0
        for o in outs:
            if isinstance(o, list) and lens:
                res.append(list1d_to_list2d(o, lens[used]))
                used += 1
            else:
                res.append(o)
        assert used in (0, len(lens)), f'{used=} != {len(lens)=}'
        return tuple(res) if len(res) > 1 else res[0]

    return wrapped
