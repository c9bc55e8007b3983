"""visual-slam_amd -- MI355X-native drop-in for visnav::matchDescriptors.

The directory name has a hyphen, so import it with
``importlib.import_module("visual-slam_amd")``.
"""
from . import synth  # noqa: F401
