import logging as _pylog

from . import import_utils  # noqa: F401


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _pylog.getLogger(name)

    @staticmethod
    def set_verbosity_info():
        _pylog.getLogger("diffusers").setLevel(_pylog.INFO)

    @staticmethod
    def set_verbosity_warning():
        _pylog.getLogger("diffusers").setLevel(_pylog.WARNING)

    @staticmethod
    def set_verbosity_error():
        _pylog.getLogger("diffusers").setLevel(_pylog.ERROR)


logging = _Logging()
