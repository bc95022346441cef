"""smirk_amd — MI355X (gfx950) implementation of SMIRK's per-frame hot path behind the reference's own class names.

    from smirk_amd import SmirkEncoder, FLAME, Renderer, SmirkGenerator      # drop-in for src.smirk_encoder / src.FLAME.FLAME /
                                                                              # src.renderer.renderer / src.smirk_generator
All arithmetic runs in hand-written HIP kernels (smirk_amd/csrc/*.hip -> lib/libsmirk_hip.so, C ABI in include/smirk_hip.h).
There is no CPU or eager-PyTorch fallback: calls raise if the library is not built or tensors are not on the GPU.
"""
from ._lib import SmirkHipError, lib  # noqa: F401
from .FLAME import FLAME  # noqa: F401
from .renderer import Renderer  # noqa: F401
from .smirk_encoder import SmirkEncoder  # noqa: F401
from .smirk_generator import SmirkGenerator  # noqa: F401
from . import masking  # noqa: F401  (drop-in for src/utils/masking.py)
from .video import VideoPipeline  # noqa: F401  (demo_video.py's frame loop, batched + streamed)

__all__ = ["FLAME", "Renderer", "SmirkEncoder", "SmirkGenerator", "SmirkHipError", "lib", "masking", "VideoPipeline"]
