class Figure:  # noqa
    def __init__(self, *a, **k):
        pass
