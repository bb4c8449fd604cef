from . import InterpolationMode, _resize as resize  # noqa: F401
