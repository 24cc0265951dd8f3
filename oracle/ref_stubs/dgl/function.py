"""`dgl.function` stand-in: only the builtin `sum` reducer token (see dgl/__init__.py)."""


class _Sum:
    def __init__(self, msg, out):
        self.msg, self.out = msg, out


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return _Sum(msg, out)
