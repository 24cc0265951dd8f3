class TestTubeLogger:
    __test__ = False

    def __init__(self, *a, **k):
        pass
