"""smirk_amd — MI355X (gfx950) implementation of SMIRK's per-frame hot path behind the reference's own class names.

    from smirk_amd import SmirkEncoder, FLAME, Renderer, SmirkGenerator      # drop-in for src.smirk_encoder / src.FLAME.FLAME /
                                                                              # src.renderer.renderer / src.smirk_generator
All arithmetic runs in hand-written HIP kernels (smirk_amd/csrc/*.hip -> lib/libsmirk_hip.so, C ABI in include/smirk_hip.h).
There is no CPU or eager-PyTorch fallback: calls raise if the library is not built or tensors are not on the GPU.
"""
import os as _os

# One hardware queue per HIP stream.  The multi-stream schedules of this package (SmirkEncoder: 3 backbone streams; OverlappedPipeline: front + generator
# streams; smirk_generator_forward: chain side streams) rely on streams running CONCURRENTLY; the HIP runtime multiplexes streams onto $GPU_MAX_HW_QUEUES
# hardware queues (default 4) and packets of one queue retire in order, so with more streams than queues a backbone stream waits for a generator kernel it does
# not depend on (measured: 8,701 -> 9,391 faces/s on the 128-frame shard).  The runtime reads the variable when it initialises (first HIP
# call of the process), so this default only takes effect if nothing touched the device before `import smirk_amd`; pipeline.OverlappedPipeline warns otherwise.
import sys as _sys

_t = _sys.modules.get("torch")
HW_QUEUES_TOO_LATE = "GPU_MAX_HW_QUEUES" not in _os.environ and _t is not None and _t.cuda.is_initialized()   # the runtime already read its default of 4
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from ._lib import SmirkHipError, lib  # noqa: E402,F401
from .FLAME import FLAME  # noqa: F401
from .renderer import Renderer  # noqa: F401
from .smirk_encoder import SmirkEncoder  # noqa: F401
from .smirk_generator import SmirkGenerator  # noqa: F401
from . import masking  # noqa: F401  (drop-in for src/utils/masking.py)
from .video import VideoPipeline  # noqa: F401  (demo_video.py's frame loop, batched + streamed)

__all__ = ["FLAME", "Renderer", "SmirkEncoder", "SmirkGenerator", "SmirkHipError", "lib", "masking", "VideoPipeline"]
