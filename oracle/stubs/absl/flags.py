class _Flags:
    def __getattr__(self, k):
        raise AttributeError(k)


FLAGS = _Flags()


def DEFINE_string(*a, **k):
    return None


DEFINE_integer = DEFINE_float = DEFINE_bool = DEFINE_boolean = DEFINE_string
