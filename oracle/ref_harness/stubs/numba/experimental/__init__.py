def jitclass(*a, **k):
    if len(a) == 1 and isinstance(a[0], type):
        return a[0]

    def wrap(cls):
        return cls
    return wrap
