"""Error paths of the C ABI (include/smirk_hip.h): every entry validates its arguments and its caller-provided workspace BEFORE it touches the device, so a
bad call returns SMIRK_ERR_BAD_ARG / SMIRK_ERR_WORKSPACE / SMIRK_ERR_UNSUPPORTED instead of launching — which is also why these checks can run in the build
container, where no GPU exists (a launch would fail differently).  Pointers below are never dereferenced on the host: 0x1000 stands for "some device address"."""
import ctypes as C

import pytest

from smirk_amd import _lib as L

OK, BAD_ARG, WORKSPACE, LAUNCH, UNSUPPORTED = 0, -1, -2, -3, -4
P = C.c_void_p(0x1000)


@pytest.fixture(scope="module")
def lib():
    return L.lib()


def test_error_strings(lib):
    for code, word in ((BAD_ARG, "argument"), (WORKSPACE, "workspace"), (LAUNCH, "launch"), (UNSUPPORTED, "support")):
        assert word in lib.smirk_strerror(code).decode().lower()
    assert lib.smirk_strerror(OK).decode()


def test_weight_gradient_rejects_bad_shapes_and_small_workspace(lib):
    need = lib.smirk_conv_wgrad_workspace_bytes(2, 8, 8, 64, 32, 3)
    assert need > 0
    f = lib.smirk_conv_wgrad_f32
    assert f(None, P, P, 2, 8, 8, 64, 32, 3, 0, P, need, None) == BAD_ARG                   # missing operand
    assert f(P, P, P, 2, 8, 8, 60, 32, 3, 0, P, need, None) == BAD_ARG                      # Cout not a multiple of 8
    assert f(P, P, P, 2, 8, 8, 64, 32, 5, 0, P, need, None) == BAD_ARG                      # 5x5 is not a layer of this network
    assert f(P, P, P, 0, 8, 8, 64, 32, 3, 0, P, need, None) == BAD_ARG                      # empty batch
    assert f(P, P, P, 2, 8, 8, 64, 32, 3, 0, P, need - 1, None) == WORKSPACE
    prev = lib.smirk_conv_wgrad_set_mode(0)
    assert lib.smirk_conv_wgrad_set_mode(-1) == 0 and lib.smirk_conv_wgrad_set_mode(-1) == prev     # the switch reports what was active and restores the default


def test_batchnorm_train_entries(lib):
    need = lib.smirk_train_reduce_workspace_bytes(64)
    fwd, bwd, cs = lib.smirk_bn_train_forward_split16, lib.smirk_bn_train_backward_split16, lib.smirk_colsum_split16
    args = lambda C_, ws: (P, 128, C_, P, P, None, 1, 1e-5, 0.1, P, P, P, P, P, P, P, ws, None)
    assert fwd(*args(60, need)) == BAD_ARG                                                  # channels must come in groups of 8
    assert fwd(*args(64, need - 8)) == WORKSPACE
    assert fwd(None, 128, 64, P, P, None, 1, 1e-5, 0.1, P, P, P, P, P, P, P, need, None) == BAD_ARG
    assert bwd(P, P, 0, 64, P, P, P, P, 1, P, P, P, P, need, None) == BAD_ARG               # no rows
    assert bwd(P, P, 128, 64, P, P, P, P, 1, P, P, P, P, 16, None) == WORKSPACE
    assert cs(P, 128, 64, None, P, need, None) == BAD_ARG
    assert cs(P, 128, 64, P, P, 0, None) == WORKSPACE


def _desc(**kw):
    d = L.SmirkConvDesc()
    base = dict(B=1, H=16, W=16, C0=32, C1=0, Cout=32, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, Ho=16, Wo=16, pad_mode=L.PAD_ZERO, act=L.ACT_NONE,
                out_mode=L.OUT_NHWC)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, v)
    return d


def test_convolution_entries(lib):
    f = lib.smirk_conv_igemm_f16x3
    assert f(None, P, None, P, None, None, None, P, None) == BAD_ARG
    assert f(_desc(), None, None, P, None, None, None, P, None) == BAD_ARG                   # no input
    assert f(_desc(C0=30), P, None, P, None, None, None, P, None) == BAD_ARG                 # split16 tensors carry 8-channel groups
    assert f(_desc(C1=32), P, None, P, None, None, None, P, None) == BAD_ARG                 # second source announced but not given
    tail = lib.smirk_conv3x3_tail_f16x3
    assert tail(_desc(Cout=64, act=L.ACT_RELU), P, None, P, P, P, P, P, P, 3, None) == BAD_ARG      # the fused tail is defined for the 32-channel last block only


def test_flame_and_renderer_reject_missing_model(lib):
    fargs = [0 if t in (L._i, L._sz) else None for t in L._SIGS["smirk_flame_forward"][1][2:]]
    assert lib.smirk_flame_forward(None, 4, *fargs) == BAD_ARG
    rargs = [0 if t in (L._i, L._sz) else None for t in L._SIGS["smirk_render_forward"][1][1:]]
    assert lib.smirk_render_forward(None, *rargs) == BAD_ARG
