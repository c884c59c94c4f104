"""Drop-in name for maua/audiovisual/patches/examples/stylegan2.py:13-68: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.patches.examples.stylegan2 import ExampleSG2Patch as _ExampleSG2Patch


class ExampleSG2Patch(_ExampleSG2Patch):
    """(defined here so that get_patch_from_file finds a patch class that belongs to this module)"""
