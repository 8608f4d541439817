"""afford-motion_amd: MI355X-native (gfx950) implementation of afford-motion's diffusion
denoising hot path.  Host code is Python on PyTorch-ROCm; all math on the path runs in
hand-written HIP kernels behind the C-ABI of include/afm_hip.h (see afm/ffi.py)."""
__version__ = "0.1.0"
