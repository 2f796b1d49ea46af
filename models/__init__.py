"""Drop-in alias of the reference's top-level `models` package: `from models import *` / `from models.nvfi import NVFi`
resolve to the MI355X-native implementation in nvfi_amd.models."""
import sys

from nvfi_amd.models import *  # noqa: F401,F403
from nvfi_amd.models import camera, nvfi, renderer, tensorf_keyframe, tensorf_model_utils, velocity_field

for _m in (camera, nvfi, renderer, tensorf_keyframe, tensorf_model_utils, velocity_field):
    sys.modules[__name__ + "." + _m.__name__.rsplit(".", 1)[1]] = _m
