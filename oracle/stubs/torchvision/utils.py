def make_grid(*a, **k):
    raise NotImplementedError


def save_image(*a, **k):
    raise NotImplementedError
