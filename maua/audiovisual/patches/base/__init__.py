"""Drop-in name for maua/audiovisual/patches/base/__init__.py:7-44: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.patches.base import MauaPatch, get_patch_from_file  # noqa: F401
