"""ctypes binding of libsmirk_hip.so (C ABI declared in include/smirk_hip.h).

There is NO fallback: if the shared object is missing or a tensor is not on the HIP device the call raises.
torch is used for device memory and streams only; every pointer handed to the library is `tensor.data_ptr()`.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libsmirk_hip.so")
LIB_PATH = os.environ.get("SMIRK_HIP_LIBRARY", LIB_PATH)      # tuning aid: A/B a differently-built libsmirk_hip.so in one gpurun
ABI_VERSION = 3

_p = C.c_void_p
_i = C.c_int
_sz = C.c_size_t


class SmirkFlameModel(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("V", "VP", "F", "n_shape", "n_exp", "KP", "n_static", "n_dyn", "n_lut",
                                         "n_full", "n_mp", "_pad")] + \
               [(n, _p) for n in ("dirs", "v_template", "lbs_weights", "jdirs", "jtemplate", "l_eyelid", "r_eyelid",
                                  "faces", "static_faces", "static_bary", "dyn_faces", "dyn_bary", "full_faces",
                                  "full_bary", "mp_faces", "mp_bary")]


class SmirkRenderMesh(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("V", "Vf", "Ff", "nnz")] + \
               [(n, _p) for n in ("keep", "faces", "nrm_ptr", "nrm_face", "nrm_corner")]


class SmirkConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "H", "W", "C0", "C1", "Cout", "KH", "KW", "stride", "pad_t", "pad_l",
                                         "Ho", "Wo", "pad_mode", "act", "out_mode")]


PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_RELU = 0, 1
OUT_NHWC, OUT_CONVT2X2 = 0, 1

_SIGS = {
    "smirk_strerror": (C.c_char_p, [_i]),
    "smirk_abi_version": (_i, []),
    "smirk_flame_workspace_bytes": (_sz, [C.POINTER(SmirkFlameModel), _i]),
    "smirk_flame_forward": (_i, [C.POINTER(SmirkFlameModel), _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                 _p, _p, _sz, _p]),
    "smirk_flame_backward_workspace_bytes": (_sz, [C.POINTER(SmirkFlameModel), _i]),
    "smirk_flame_backward": (_i, [C.POINTER(SmirkFlameModel), _p, _i, _p, _i, _p, _i] + [_p] * 5 + [_p, _p] + [_p] * 4 + [_p] * 7 +
                             [_p, _sz, _p]),
    "smirk_vertices2landmarks": (_i, [_p, _i, _i, _p, _p, _p, _i, _p, _p]),
    "smirk_render_workspace_bytes": (_sz, [C.POINTER(SmirkRenderMesh), _i, _i, _i]),
    "smirk_render_forward": (_i, [C.POINTER(SmirkRenderMesh), _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "smirk_vertex_normals": (_i, [C.POINTER(SmirkRenderMesh), _i, _p, _p, _p]),
    "smirk_mask_face_weights": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "smirk_sample_faces": (_i, [_p, _i, _i, _i, C.c_uint64, C.c_uint64, _p, _p, _p]),
    "smirk_points_to_pixels": (_i, [_p, _i, _i, _i, _p, _p]),
    "smirk_maxpool_sq": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "smirk_bernoulli_field": (_i, [_p, _sz, C.c_float, C.c_uint64, C.c_uint64, _p]),
    "smirk_masking_compose": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, C.c_uint64, C.c_uint64, _p, _p]),
    "smirk_render_backward_workspace_bytes": (_sz, [C.POINTER(SmirkRenderMesh), _i, _i, _i]),
    "smirk_render_backward": (_i, [C.POINTER(SmirkRenderMesh), _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "smirk_project_landmarks_backward": (_i, [_p, _p, _p, _i, _i, _p, _p, _p]),
    "smirk_mbconv_lds_bytes": (_sz, [_i, _i, _i, _i]),
    "smirk_mbconv_supported": (_i, [_i, _i, _i, _i]),
    "smirk_mbconv_fused_split16": (_i, [_p] * 10 + [_i, _p] + [_i] * 7 + [_p]),
    "smirk_warp_affine_u8": (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _p, _p, _p]),
    "smirk_resize_linear_u8": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "smirk_f32_nchw_to_u8_grid": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _p]),
    "smirk_u8_hwc_to_f32_nchw": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "smirk_interp_bilinear_f32": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "smirk_hull_mask": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "smirk_rendered_mask": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "smirk_scatter_points_mask": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "smirk_transfer_pixels": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "smirk_project_landmarks": (_i, [_p, _p, _i, _i, _p, _p]),
    "smirk_conv_igemm_f32": (_i, [C.POINTER(SmirkConvDesc), _p, _p, _p, _p, _p, _p, _p, _p]),
    "smirk_conv_igemm_f16x3": (_i, [C.POINTER(SmirkConvDesc), _p, _p, _p, _p, _p, _p, _p, _p]),
    "smirk_conv3x3_tail_f16x3": (_i, [C.POINTER(SmirkConvDesc), _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "smirk_f32_to_split16": (_i, [_p, _p, _sz, _p]),
    "smirk_split16_to_f32": (_i, [_p, _p, _sz, _p]),
    "smirk_maxpool2x2_split16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "smirk_pack_generator_input_split16": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _p]),
    "smirk_conv1x1_sigmoid_nchw_split16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "smirk_maxpool2x2_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "smirk_nchw_to_nhwc_pad": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "smirk_pack_generator_input": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "smirk_conv1x1_sigmoid_nchw": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "smirk_stem_conv_s2": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "smirk_dwconv3x3": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smirk_gap_linear": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "smirk_stem_conv_s2_split16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "smirk_dwconv3x3_split16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smirk_gap_linear_split16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "smirk_expression_clamps": (_i, [_p, _i, _i, _p]),
}
EXPORTS = tuple(_SIGS)

_LIB = None


class SmirkHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SmirkHipError(f"{LIB_PATH} not found: build it with `python -m smirk_amd.build` "
                                "(smirk_amd has no CPU / eager fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)           # AttributeError => header / library mismatch
            fn.restype, fn.argtypes = res, args
        if L.smirk_abi_version() != ABI_VERSION:
            raise SmirkHipError("libsmirk_hip.so ABI version mismatch; rebuild")
        _LIB = _TimedLib(L)
    return _LIB


def check(code):
    if code != 0:
        raise SmirkHipError(f"libsmirk_hip: {lib().smirk_strerror(code).decode()} ({code})")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32, allow_none=False):
    """Device pointer of a contiguous tensor of the expected dtype that lives on the HIP device."""
    if t is None:
        if allow_none:
            return None
        raise SmirkHipError("missing tensor argument")
    if not t.is_cuda:
        raise SmirkHipError("smirk_amd runs on the MI355X HIP device only: got a CPU tensor (no CPU fallback exists)")
    if t.dtype != dtype:
        raise SmirkHipError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise SmirkHipError("expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


def as_f32c(t):
    """float32 contiguous view/copy (device-side plumbing only)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# bench.py sets TIMER to a list to collect per-launch (kernel, algorithmic flop, algorithmic bytes, start, end) HIP events on the launch stream
TIMER = None
_IN_TIMED = False


def timed(kernel, flops, fn, nbytes=None):
    """Bracket one library launch with HIP events on the current (= launch) stream when bench.py has armed TIMER."""
    global _IN_TIMED
    if TIMER is None or _IN_TIMED:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _IN_TIMED = True
    try:
        e0.record()
        r = fn()
        e1.record()
    finally:
        _IN_TIMED = False
    TIMER.append((kernel, flops, nbytes, e0, e1))
    return r


class _TimedLib:
    """Attribute proxy over the CDLL: with TIMER armed, every entry point that enqueues work is timed under its own name (launches that a
    caller already wrapped in timed() keep the caller's more specific label)."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if TIMER is None or _IN_TIMED or name.endswith("_bytes") or name in ("smirk_strerror", "smirk_abi_version", "smirk_mbconv_supported"):
            return fn
        return lambda *a: timed(name, None, lambda: fn(*a))


def igemm_kernel_name(n_gemm, split=False, c0=32, c1=0, k=3):
    """Which conv_igemm_kernel instantiation smirk_conv_igemm_{f32,f16x3} dispatches (mirrors launch_igemm in conv.hip): tile by GEMM N,
    K-walk mode by channel counts (0 generic, 1 tap-major pointer walk, 2 1x1 with a partial chunk, 4 buffer-addressed tap-major,
    5 buffer-addressed channel-major).  Used for labelling timings only."""
    t = "128,128,2,2" if n_gemm > 64 else "128,64,2,2" if n_gemm > 32 else "256,32,4,1"
    aligned = c0 % 32 == 0 and c1 % 32 == 0
    kw = 1 if aligned else (2 if k == 1 else 0)
    if split and aligned and os.environ.get("SMIRK_IGEMM_LEAN", "1") != "0":
        pow2 = (c0 & (c0 - 1)) == 0 and (c1 & (c1 - 1)) == 0
        kw = 5 if (k == 3 and pow2 and ((c0 + c1) // 32) % 2 == 0 and n_gemm > 64) else 4
    return f"conv_igemm_kernel<{t},{'true' if split else 'false'},{kw}>"


class Workspace:
    """Grow-only byte scratch buffer owned by a module (the library never allocates)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return self.buf
