def overload(*a, **k):
    def wrap(fn):
        return fn
    return wrap
