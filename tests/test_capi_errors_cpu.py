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
    args = lambda C_, ws: (P, 128, C_, P, P, None, 1, 1e-5, 0.1, P, P, None, P, P, P, P, P, ws, None)
    assert fwd(*args(60, need)) == BAD_ARG                                                  # channels must come in groups of 8
    assert fwd(*args(64, need - 8)) == WORKSPACE
    assert fwd(None, 128, 64, P, P, None, 1, 1e-5, 0.1, P, P, None, P, P, P, P, P, need, None) == BAD_ARG
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


def test_fused_encoder_entries_round3(lib):
    """encoder_head.hip / mbconv_image.hip: what the `*_supported` predicates admit is exactly the blocks of the two backbones (smirk_encoder.py:14-110 via timm's
    tf_mobilenetv3_{small,large}_minimal_100), and the launch entries refuse everything else before touching the device."""
    hs = lib.smirk_encoder_head_supported
    assert hs(16, 0, 16, 16, 16, 1, 1) == 1 and hs(16, 0, 16, 16, 16, 2, 0) == 1          # large: stride-1 block with skip; small: stride-2 block
    assert hs(16, 0, 16, 16, 16, 2, 1) == 0                                               # a residual cannot cross a stride
    assert hs(32, 0, 16, 16, 16, 1, 0) == 0 and hs(16, 1, 16, 16, 16, 1, 0) == 0 and hs(16, 0, 16, 16, 24, 1, 0) == 0
    head = lib.smirk_encoder_head_fused_split16
    ok = [P] * 10 + [0, P, 4, 224, 224, 1, None]
    bad = lambda i, v: [v if j == i else a for j, a in enumerate(ok)]
    for i in list(range(10)) + [11]:
        assert head(*bad(i, None)) == BAD_ARG                                             # every operand is required
    assert head(*bad(12, 0)) == BAD_ARG and head(*bad(13, 2)) == BAD_ARG and head(*bad(15, 3)) == BAD_ARG
    assert head(*[P] * 10, 1, P, 4, 224, 224, 2, None) == BAD_ARG                         # residual with stride 2

    ms = lib.smirk_mbconv_image_supported
    for H, Cin, mid, Cout in ((14, 80, 200, 80), (14, 80, 184, 80), (14, 80, 480, 112), (14, 112, 672, 112),       # large, 14 x 14
                              (14, 40, 240, 40), (14, 40, 120, 48), (14, 48, 144, 48), (7, 96, 576, 96)):           # small, 14 x 14 and 7 x 7
        assert ms(H, H, Cin, mid, Cout, 1) == 1, (H, Cin, mid, Cout)                     # (round 5: the 40 / 48-channel blocks too, K padded to a multiple of 16)
    for H, Cin, mid, Cout in ((56, 24, 72, 24), (28, 40, 120, 40), (28, 24, 88, 24), (112, 24, 72, 24)):           # round 5: 14 x 14 halo tiles of larger images
        assert ms(H, H, Cin, mid, Cout, 1) == 1, (H, Cin, mid, Cout)
    assert ms(30, 30, 40, 120, 40, 1) == 0 and ms(28, 30, 40, 120, 40, 1) == 0            # ... whose sides are multiples of 14
    assert ms(28, 28, 40, 120, 40, 2) == 0
    assert ms(14, 14, 80, 200, 80, 2) == 0                                                # stride-2 blocks are not whole-image blocks
    assert ms(28, 28, 80, 200, 80, 1) == 0                                                # an image must fit the 224-row workgroup
    assert ms(7, 7, 160, 960, 160, 1) == 0                                                # the 160-channel stage is not instantiated
    img = lib.smirk_mbconv_image_split16
    ok = [P] * 10 + [1, P, 8, 14, 14, 80, 200, 80, None]
    for i in list(range(10)) + [11]:
        assert img(*[None if j == i else a for j, a in enumerate(ok)]) == BAD_ARG
    assert img(*[P] * 10, 1, P, 8, 14, 14, 80, 480, 112, None) == BAD_ARG                 # residual needs Cin == Cout
    assert img(*[P] * 10, 0, P, 8, 28, 28, 80, 200, 80, None) == UNSUPPORTED
    assert img(*[P] * 10, 0, P, 8, 14, 14, 72, 200, 80, None) == UNSUPPORTED              # channels in groups of 16
    assert img(*[P] * 10, 0, P, 0, 14, 14, 80, 200, 80, None) == BAD_ARG


def test_eval_batchnorm_and_batched_packing_entries_round3(lib):
    f, b = lib.smirk_bn_eval_forward_split16, lib.smirk_bn_eval_backward_split16
    ok = [P, 128, 64, P, P, P, P, None, 1, 1e-5, P, P, P, None]
    assert f(*[None if j == 0 else a for j, a in enumerate(ok)]) == BAD_ARG
    assert f(*[60 if j == 2 else a for j, a in enumerate(ok)]) == BAD_ARG                 # channels in groups of 8
    assert f(*[0 if j == 1 else a for j, a in enumerate(ok)]) == BAD_ARG                  # no rows
    assert f(*[4096 if j == 2 else a for j, a in enumerate(ok)]) == BAD_ARG               # more 8-channel groups than a workgroup has lanes
    for j in (3, 4, 5, 6, 10, 11, 12):
        assert f(*[None if k == j else a for k, a in enumerate(ok)]) == BAD_ARG
    okb = [P, P, 128, 64, P, P, P, P, P, 1, P, None]
    for j in (0, 1, 4, 5, 6, 7, 8, 10):
        assert b(*[None if k == j else a for k, a in enumerate(okb)]) == BAD_ARG
    assert b(*[0 if k == 2 else a for k, a in enumerate(okb)]) == BAD_ARG and b(*[12 if k == 3 else a for k, a in enumerate(okb)]) == BAD_ARG
    pk = lib.smirk_pack_conv_weights_batch_split16
    assert pk(None, 4, 100, None) == BAD_ARG and pk(P, 0, 100, None) == BAD_ARG and pk(P, 4097, 100, None) == BAD_ARG and pk(P, 4, 0, None) == BAD_ARG


def test_maxpool_entry_round3(lib):
    mp = lib.smirk_maxpool_sq
    assert mp(None, P, P, 1, 224, 224, 10, 0, None) == BAD_ARG and mp(P, P, None, 1, 224, 224, 10, 0, None) == BAD_ARG
    assert mp(P, P, P, 0, 224, 224, 10, 0, None) == BAD_ARG and mp(P, P, P, 1, 224, 224, -1, 0, None) == BAD_ARG
