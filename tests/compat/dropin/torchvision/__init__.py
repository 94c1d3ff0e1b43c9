"""Minimal `torchvision` name shim for the reference scripts (`inference.py:21,33,86-94,417-419`): ToTensor / Normalize / Compose
and utils.save_image, nothing else.  Test / demo environment only (no torchvision wheel in this image)."""
from . import transforms, utils  # noqa: F401

__version__ = "0.0-idmvton-shim"
