"""Mirror of the reference's `models` package surface for the hot path (models/__init__.py:1-6)."""
from .camera import Ray, Camera, BatchedRays
from .renderer import Renderer
from .nvfi import NVFi
from .tensorf_model_utils import AlphaGridMask
from .mask_field import MaskField
from .velocity_field import VelBasis, VelocityAABB, VelocityAABBSur, N_to_reso
from .tensorf_keyframe import TensorVMKeyframeTimeKplane, DeviceTime
