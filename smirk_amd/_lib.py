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
ABI_VERSION = 11
PACK_DEPTHWISE, PACK_STEM, PACK_CONVT2X2 = -3, -27, -2          # SmirkPackJob.KH markers (include/smirk_hip.h SMIRK_PACK_*)
SMIRK_OK, SMIRK_ERR_BAD_ARG, SMIRK_ERR_WORKSPACE, SMIRK_ERR_LAUNCH, SMIRK_ERR_UNSUPPORTED = 0, -1, -2, -3, -4      # include/smirk_hip.h

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


class SmirkConvLayer(C.Structure):
    _fields_ = [("w", _p), ("scale", _p), ("shift", _p)]


GEN_MAX_RES, BACKBONE_MAX_BLOCKS = 16, 24
PRECISION_F32, PRECISION_F16X3 = 0, 1


class SmirkGeneratorWeights(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_channels", "out_channels", "features", "res_blocks", "precision", "cin_pad")] + \
               [("enc", SmirkConvLayer * 2 * 5), ("res", SmirkConvLayer * 2 * GEN_MAX_RES), ("up", SmirkConvLayer * 4),
                ("dec", SmirkConvLayer * 2 * 4), ("final_w", _p), ("final_b", _p)]


class SmirkMbBlock(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "stride", "cin", "mid", "cout", "skip")] + \
               [("pw", SmirkConvLayer), ("dw", SmirkConvLayer), ("pwl", SmirkConvLayer)]


class SmirkBackboneWeights(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_blocks", "precision", "n_out", "clamp_n_exp", "stem_cout", "feat_ch")] + \
               [("stem", SmirkConvLayer), ("blocks", SmirkMbBlock * BACKBONE_MAX_BLOCKS), ("head_w", _p), ("head_b", _p)]


class SmirkProfileRecord(C.Structure):
    _fields_ = [("kernel", C.c_char * 120), ("flop", C.c_double), ("bytes", C.c_double), ("ms", C.c_float), ("_pad", C.c_int32)]


PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_RELU = 0, 1
OUT_NHWC, OUT_CONVT2X2 = 0, 1

_SIGS = {
    "smirk_strerror": (C.c_char_p, [_i]),
    "smirk_abi_version": (_i, []),
    "smirk_range_flag_peek": (C.c_uint, []),
    "smirk_range_flag_clear": (None, []),
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
    "smirk_encoder_head_supported": (_i, [_i] * 7),
    "smirk_encoder_head_fused_split16": (_i, [_p] * 10 + [_i, _p] + [_i] * 4 + [_p]),
    "smirk_mbconv_image_supported": (_i, [_i] * 6),
    "smirk_mbconv_image_split16": (_i, [_p] * 10 + [_i, _p] + [_i] * 6 + [_p]),
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
    "smirk_conv_igemm_f16x1": (_i, [C.POINTER(SmirkConvDesc), _p, _p, _p, _p, _p, _p, _p, _p]),
    "smirk_conv3x3_tail_f16x3": (_i, [C.POINTER(SmirkConvDesc), _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "smirk_conv3x3_pool_f16x3": (_i, [C.POINTER(SmirkConvDesc), _p, _p, _p, _p, _p, _p, _p, _p]),
    "smirk_enc1_fused_supported": (_i, [_i, _i, _i, _i]),
    "smirk_enc1_fused_split16": (_i, [_p] * 9 + [_i, _i, _i, _p]),
    "smirk_f32_to_split16": (_i, [_p, _p, _sz, _p]),
    "smirk_split16_to_f32": (_i, [_p, _p, _sz, _p]),
    "smirk_maxpool2x2_split16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "smirk_split16_range_check": (_i, [_p, _sz, C.c_float, _p, _p]),
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
    "smirk_generator_workspace_bytes": (_sz, [C.POINTER(SmirkGeneratorWeights), _i, _i, _i]),
    "smirk_generator_forward": (_i, [C.POINTER(SmirkGeneratorWeights), _p, _i, _p, _i, _p, _i, _i, _i, C.POINTER(_p), _p, _sz, _p]),
    "smirk_backbone_workspace_bytes": (_sz, [C.POINTER(SmirkBackboneWeights), _i, _i, _i]),
    "smirk_backbone_forward": (_i, [C.POINTER(SmirkBackboneWeights), _p, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "smirk_train_reduce_workspace_bytes": (_sz, [_i]),
    "smirk_conv_stats_rows_max": (_sz, [C.POINTER(SmirkConvDesc)]),
    "smirk_conv_igemm_stats_split16": (_i, [C.POINTER(SmirkConvDesc), _p, _p, _p, _p, _p, C.POINTER(C.c_int), _i, _p]),
    "smirk_bn_train_forward_partials_split16": (_i, [_p, _sz, _i, _p, _p, _p, _i, C.c_float, C.c_float, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "smirk_dwconv3x3_stats_split16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, C.POINTER(C.c_int), _p]),
    "smirk_bn_train_forward_split16": (_i, [_p, _sz, _i, _p, _p, _p, _i, C.c_float, C.c_float, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "smirk_bn_train_backward_split16": (_i, [_p, _p, _sz, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _sz, _p]),
    "smirk_bn_eval_forward_split16": (_i, [_p, _sz, _i, _p, _p, _p, _p, _p, _i, C.c_float, _p, _p, _p, _p]),
    "smirk_bn_eval_backward_split16": (_i, [_p, _p, _sz, _i, _p, _p, _p, _p, _p, _i, _p, _p]),
    "smirk_colsum_split16": (_i, [_p, _sz, _i, _p, _p, _sz, _p]),
    "smirk_maxpool2x2_backward_split16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "smirk_reflect_pad1_backward_split16": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "smirk_space_to_depth2_split16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "smirk_conv1x1_sigmoid_backward_split16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "smirk_conv_wgrad_set_mode": (_i, [_i]),
    "smirk_conv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "smirk_conv_wgrad_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "smirk_conv_wgrad_f16x1": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "smirk_conv_wgrad_param": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "smirk_conv_wgrad_x1_fallbacks": (C.c_ulonglong, []),
    "smirk_pack_conv_weights_batch_split16": (_i, [_p, _i, C.c_ulonglong, _p]),
    "smirk_pack_conv_weights_split16": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "smirk_stem_conv_s2_raw_split16": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "smirk_stem_conv_s2_wgrad_workspace_bytes": (_sz, [_i]),
    "smirk_stem_conv_s2_wgrad_split16": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "smirk_stem_conv_s2_dgrad_split16": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "smirk_dwconv3x3_dgrad_split16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "smirk_dwconv3x3_wgrad_workspace_bytes": (_sz, [_i]),
    "smirk_dwconv3x3_wgrad_split16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "smirk_dwconv3x3_wgrad_param_split16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "smirk_gap_linear_backward_split16": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "smirk_profile_start": (_i, []),
    "smirk_profile_stop": (_i, [C.POINTER(SmirkProfileRecord), _i]),
    "smirk_random_point_budget": (_i, [_p, _i, _i, C.c_float, C.c_uint64, C.c_uint64, _p]),
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
        _LIB = L
    return _LIB


def check(code):
    if code != 0:
        raise SmirkHipError(f"libsmirk_hip: {lib().smirk_strerror(code).decode()} ({code})")


def raise_if_range_tripped(where, synchronize=False):
    """The split-fp16 storage format carries |x| < 65520 only (its `hi` half is an fp16).  Every kernel that writes it audits what it stores and sets one sticky
    host-pinned word (include/smirk_hip.h smirk_range_flag_peek); this reads that word — a plain host load, no device synchronisation unless asked for — and
    raises once it is set.  The modules call it at the START of every forward, so an overflow in call N is reported by call N + 1 at the latest (or by
    smirk_amd.check_numerics(), which synchronises first).  The reference computes in fp32 and has no such limit: a checkpoint whose activations leave the
    fp16 range must run SmirkGenerator.precision = "f32" (exact-fp32 kernels) instead."""
    if synchronize:
        torch.cuda.synchronize()
    L = lib()
    if L.smirk_range_flag_peek():
        L.smirk_range_flag_clear()
        raise SmirkHipError(f"{where}: an activation left the range of the split-fp16 storage format (|x| >= 65520, or a non-finite input) in a kernel that ran "
                            "before this call — its output and everything computed from it are invalid (inf / NaN).  Badly scaled weights or inputs; the exact-fp32 "
                            "generator kernels (SmirkGenerator.precision = 'f32') have no such limit.  The flag has been cleared.")


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


def profile_start():
    """Arm the library's launch profiler (include/smirk_hip.h): every kernel it launches from now on is bracketed by HIP events on its stream."""
    check(lib().smirk_profile_start())


def profile_stop(cap=8192):
    """-> [(kernel name as instantiated, algorithmic flop, algorithmic bytes, milliseconds)] for every launch since profile_start()."""
    buf = (SmirkProfileRecord * cap)()
    n = lib().smirk_profile_stop(buf, cap)
    if n < 0:
        check(n)
    return [(buf[i].kernel.decode(), buf[i].flop, buf[i].bytes, buf[i].ms) for i in range(min(n, cap))]


def conv_layer(w, scale=None, shift=None):
    """SmirkConvLayer from packed device tensors (the caller keeps them alive)."""
    l = SmirkConvLayer()
    l.w = w.data_ptr() if w is not None else None
    l.scale = scale.data_ptr() if scale is not None else None
    l.shift = shift.data_ptr() if shift is not None else None
    return l


class SmirkPackJob(C.Structure):
    """include/smirk_hip.h SmirkPackJob"""
    _fields_ = [("w", C.c_void_p), ("fwd", C.c_void_p), ("dgrad", C.c_void_p), ("Cout", C.c_int32), ("cin_total", C.c_int32), ("cin_off", C.c_int32),
                ("Cin", C.c_int32), ("KH", C.c_int32), ("cin_pad", C.c_int32), ("start", C.c_ulonglong)]


class _LoudCut(torch.autograd.Function):
    """Identity whose backward raises: attached to the output of a forward that has no backward implementation whenever autograd would
    otherwise record a (silently broken) graph through it.  Forward-only callers (demo.py runs without torch.no_grad()) are unaffected."""

    @staticmethod
    def forward(ctx, what, y, *deps):
        ctx.what = what
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError(f"{ctx.what} has no backward on the HIP path: gradients cannot flow through it "
                                  "(run it under torch.no_grad(), or detach its inputs)")


def loud_cut(what, y, deps):
    """y if autograd is off or nothing in `deps` requires grad; otherwise y with a grad_fn that raises when back-propagated through."""
    if not torch.is_grad_enabled():
        return y
    deps = [t for t in deps if torch.is_tensor(t) and t.requires_grad]
    return _LoudCut.apply(what, y, *deps) if deps else y


class Workspace:
    """Grow-only byte scratch buffers owned by a module, one per (device, stream): the library never allocates, and two streams that run
    the same module concurrently (OverlappedPipeline, the encoder's side streams) must not share scratch memory."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes, device, stream=None):
        if torch.device(device).type != "cuda":
            raise SmirkHipError("smirk_amd runs on the MI355X HIP device only: got a CPU tensor / module (no CPU fallback exists)")
        key = (device, torch.cuda.current_stream(device).cuda_stream if stream is None else stream)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self.bufs[key] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return buf
