class AverageMeter:
    def __init__(self):
        self.sum = 0.
        self.count = 0
        self.avg = 0.

    def update(self, v, n=1):
        self.sum += float(v) * n
        self.count += n
        self.avg = self.sum / max(self.count, 1)


def load_obj(f):
    import pickle
    return pickle.load(open(f, 'rb'))


def dump_obj(o, f):
    import pickle
    pickle.dump(o, open(f, 'wb'))


def glob_all(*a, **k):
    return []


def sort_file_by_time(x):
    return x
