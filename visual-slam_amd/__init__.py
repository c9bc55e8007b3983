"""visual-slam_amd -- MI355X-native drop-in for visnav::matchDescriptors
(/root/reference/include/visnav/keypoints.h:247-361).

The directory name has a hyphen, so import it with
``importlib.import_module("visual-slam_amd")``.

  matcher.match_descriptors / matcher.Matcher   the path, through the C ABI
  synth                                         deterministic synthetic descriptor sets
  sfm_pairs                                     pair enumeration + multi-GPU sharding
"""
from . import synth  # noqa: F401
from . import matcher  # noqa: F401
from .matcher import Matcher, MultiMatcher, VsmError, match_descriptors  # noqa: F401
