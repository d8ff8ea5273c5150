def rotate(*a, **k):
    raise NotImplementedError
