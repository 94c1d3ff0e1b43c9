"""`from ip_adapter.ip_adapter import Resampler` (train_xl.py:45; reference ip_adapter/ip_adapter.py:26 re-exports it)."""
from .resampler import Resampler  # noqa: F401
