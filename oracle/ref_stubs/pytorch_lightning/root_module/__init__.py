from . import root_module  # noqa: F401
