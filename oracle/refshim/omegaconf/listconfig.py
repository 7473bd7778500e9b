from . import ListConfig  # noqa: F401
