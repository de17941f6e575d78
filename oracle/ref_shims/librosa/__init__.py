"""Shim for the three librosa 0.6.0 entry points the reference touches."""
from . import filters, util  # noqa
