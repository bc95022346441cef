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
# not depend on (measured: 8,701 -> 9,391 faces/s on the 128-frame shard).
#
# The runtime reads the variable ONCE, on the first HIP call of the process — any HIP call: torch.cuda.init(), a tensor on the device, but also the
# hipGetDeviceCount behind torch.cuda.is_available() / device_count(), which leave torch.cuda.is_initialized() False.  The only reliable way to set it is to
# export GPU_MAX_HW_QUEUES=16 before the process starts (INTEGRATION.md says so); the default set here is a convenience that works when `import smirk_amd`
# comes before the first HIP call.  It changes a process-wide variable (other HIP libraries in the process see it too) and never overrides an exported value.
# HW_QUEUES_STATE says what is known:  "exported"  the user set it; "default-in-time"  set here and torch had not been imported yet; "default-after-torch"
# set here after `import torch` — in time unless something already made a HIP call (cannot be observed from Python); "too-late"  torch reports an
# initialised device or a cached device count, the runtime runs on its default of 4.  Results never depend on it (tests/test_dropin_gpu.py), only overlap.
import sys as _sys

_t = _sys.modules.get("torch")


def _hip_touched(t):
    try:
        return bool(t.cuda.is_initialized() or getattr(t.cuda, "_cached_device_count", None) is not None)
    except Exception:                                               # noqa: BLE001
        return False


if "GPU_MAX_HW_QUEUES" in _os.environ:
    HW_QUEUES_STATE = "exported"
elif _t is None:
    HW_QUEUES_STATE = "default-in-time"
else:
    HW_QUEUES_STATE = "too-late" if _hip_touched(_t) else "default-after-torch"
HW_QUEUES_TOO_LATE = HW_QUEUES_STATE == "too-late"                 # the runtime already read its default of 4
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from ._lib import SmirkHipError, lib  # noqa: E402,F401
from .FLAME import FLAME  # noqa: F401
from .renderer import Renderer  # noqa: F401
from .smirk_encoder import SmirkEncoder  # noqa: F401
from .smirk_generator import SmirkGenerator  # noqa: F401
from . import masking  # noqa: F401  (drop-in for src/utils/masking.py)
from .video import VideoPipeline  # noqa: F401  (demo_video.py's frame loop, batched + streamed)



def check_numerics():
    """Synchronise the device and raise SmirkHipError if any kernel of this library stored a value that the split-fp16 activation format cannot carry
    (|x| >= 65520 or a non-finite input) since the last check.  The modules perform the same check — without synchronising — at the start of every forward, so an
    overflow is reported by the next call at the latest; call this after the LAST forward of a job (e.g. before writing results to disk)."""
    from ._lib import raise_if_range_tripped
    raise_if_range_tripped("smirk_amd.check_numerics", synchronize=True)


__all__ = ["FLAME", "Renderer", "SmirkEncoder", "SmirkGenerator", "SmirkHipError", "lib", "masking", "VideoPipeline", "check_numerics"]
