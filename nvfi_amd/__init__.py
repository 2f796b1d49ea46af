"""nvfi_amd - MI355X-native NVFi hot path (hand-written HIP for gfx950 behind a C ABI).

`nvfi_amd.models` mirrors the reference's `models` package surface (NVFi, Renderer, Camera, Ray,
AlphaGridMask, VelBasis, ...) for the render + physics-loss hot path; everything between
"rays in" and "rgb/depth/loss + gradients out" runs in libnvfi_hip.so.
"""
__version__ = "0.1.0"
