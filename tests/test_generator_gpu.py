"""GPU parity: smirk_amd.SmirkGenerator (fp32 MFMA implicit-GEMM convs) vs the torch-CPU fp32 oracle and the committed
reference output.  Tolerance: fp32 accumulation-order differences over 32 conv layers."""
import os

import numpy as np
import pytest
import torch

from oracle import assets as A
from oracle import generator_ref as G

pytestmark = pytest.mark.gpu

OUT_TOL = 2e-5        # abs, on the sigmoid output in [0,1]
ACT_RTOL = 2e-4       # intermediate activations, relative to the tensor's max magnitude


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def gen(request):
    """both arithmetic modes must meet the SAME tolerances: split-fp16 (default) and the exact fp32 FMA-chain path"""
    from smirk_amd import SmirkGenerator
    sd = G.synth_state_dict()
    m = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
    m.load_state_dict(sd, strict=True)
    m.precision = request.param
    return m.cuda().eval(), sd


def test_generator_matches_reference_golden(gen, golden_dir):
    m, sd = gen
    g = np.load(os.path.join(golden_dir, "generator_golden.npz"))
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) - float(g["w_checksum"])) < 1e-6 * float(g["w_checksum"])
    x = A.synth_generator_input(1, seed=int(g["seed"]))
    with torch.no_grad():
        y = m(x.cuda()).cpu().numpy()
    assert np.abs(y[:, :, ::4, ::4] - g["y_sub4"]).max() < OUT_TOL
    assert abs(y.astype(np.float64).sum() - float(g["y_sum"])) < 1e-5 * float(g["y_sum"])


def test_generator_layerwise_vs_oracle(gen):
    m, sd = gen
    x = A.synth_generator_input(2, seed=3)
    rt, gt = {}, {}
    yr = G.forward(sd, x, taps=rt)
    yg = m(x.cuda(), _taps=gt)
    torch.cuda.synchronize()
    for k in ("enc1", "enc2", "enc3", "enc4", "bottleneck", "res", "dec4", "dec3", "dec2", "dec1"):
        from smirk_amd.smirk_generator import split16_to_float
        act = split16_to_float(gt[k]) if m.precision == "f16x3" else gt[k]
        a, b = act.permute(0, 3, 1, 2).cpu(), rt[k]
        err = (a - b).abs().max().item() / b.abs().max().item()
        assert err < ACT_RTOL, (k, err)
    assert (yg.cpu() - yr).abs().max().item() < OUT_TOL


def test_generator_forward_pair_equals_cat(gen):
    m, _ = gen
    x = A.synth_generator_input(2, seed=8).cuda()
    y0 = m(x)
    y1 = m.forward_pair(x[:, :3].contiguous(), x[:, 3:].contiguous())            # two-source entry: the cat is never materialised
    assert torch.equal(y0, y1)


def test_generator_odd_batch_and_small_image(gen):
    """ragged sizes: B not a multiple of any tile, and a 32x32 image (bottleneck 2x2 => reflect padding of a 2-wide map)."""
    m, sd = gen
    x = A.synth_generator_input(3, seed=1)[:, :, 96:128, 64:96].contiguous()
    yr = G.forward(sd, x)
    yg = m(x.cuda()).cpu()
    assert (yg - yr).abs().max().item() < OUT_TOL


def test_generator_train_mode_rejects_bad_shapes_loudly(gen):
    """train mode is the HIP training path (tests/test_generator_train_gpu.py); what it cannot run must raise, not fall back"""
    from smirk_amd import SmirkHipError
    m, _ = gen
    m.train()
    try:
        with pytest.raises(SmirkHipError):
            m(torch.zeros(1, 6, 40, 40).cuda())                    # H, W not multiples of 16
        with pytest.raises(SmirkHipError):
            m(torch.zeros(1, 5, 32, 32).cuda())                    # wrong channel count
    finally:
        m.eval()


def test_f16x3_range_check_flags_a_badly_scaled_checkpoint(monkeypatch):
    """$SMIRK_F16X3_RANGE_CHECK: activations beyond the fp16 range of the split format must be reported per block, not propagate as inf/NaN."""
    from smirk_amd import SmirkGenerator, SmirkHipError
    sd = G.synth_state_dict()
    m = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = A.synth_generator_input(1, seed=5).cuda()
    monkeypatch.setenv("SMIRK_F16X3_RANGE_CHECK", "1")
    with torch.no_grad():
        y = m(x)                                                   # a calibrated network passes the audit
        assert torch.isfinite(y).all()
        m.encoder3.enc3norm2.weight.mul_(3e5)                      # blow one block's activations past 65504
        with pytest.raises(SmirkHipError, match="enc3"):
            m(x)
        import smirk_amd
        with pytest.raises(SmirkHipError, match="split-fp16"):     # the ALWAYS-ON flag (no environment switch) saw the same overflow; reading it clears it
            smirk_amd.check_numerics()
        m.precision = "f32"                                        # the exact-fp32 mode carries them (no audit needed)
        assert torch.isfinite(m(x)).all()
        smirk_amd.check_numerics()                                 # ... and does not trip the flag


def test_split_fp16_overflow_raises_instead_of_returning_nan_pixels():
    """the only silent-wrong-answer mode of the library.  A deliberately 1e4-scaled state_dict (every conv weight x 1e4: each weight still fits
    the format, the activations of the second layer do not) is loaded with NO environment switch set: the forward that overflows returns (nothing synchronises on
    the hot path), and the NEXT call — or smirk_amd.check_numerics() — raises SmirkHipError; the flag is sticky until read, healthy weights do not trip it, and a
    NaN in the input image is reported the same way."""
    import smirk_amd
    from smirk_amd import SmirkGenerator, SmirkHipError
    good = G.synth_state_dict()
    bad = {k: (v * 1e4 if (k.endswith("conv1.weight") or k.endswith("conv2.weight")) else v) for k, v in good.items()}
    m = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
    m.load_state_dict(bad)
    m = m.cuda().eval()
    x = A.synth_generator_input(2, seed=5).cuda()
    with torch.no_grad():
        y = m(x)                                                   # overflows on the device; returns without synchronising
        torch.cuda.synchronize()
        assert not bool(((y > 1e-3) & (y < 1 - 1e-3)).all())       # (the pixels ARE garbage: saturated / NaN — exactly what must not pass silently)
        with pytest.raises(SmirkHipError, match="split-fp16"):
            m(x)                                                   # "raise on the next call"
        y = m(x)                                                   # the raise cleared the flag: this call runs, and trips it again
        with pytest.raises(SmirkHipError, match="split-fp16"):
            smirk_amd.check_numerics()                             # explicit form: synchronises, then reads the flag
        m.load_state_dict(good)
        y = m(x)
        smirk_amd.check_numerics()                                 # healthy weights: silent
        assert torch.isfinite(y).all()
        xn = x.clone(); xn[1, 2, 100, 100] = float("nan")
        m(xn)
        with pytest.raises(SmirkHipError, match="split-fp16"):
            smirk_amd.check_numerics()
        # a weight the format cannot carry at all is refused when the weight image is built, naming the layer
        worse = dict(good); worse["encoder2.enc2conv1.weight"] = good["encoder2.enc2conv1.weight"] * 1e7
        m.load_state_dict(worse)
        with pytest.raises(SmirkHipError, match="enc2"):
            m(x)


@pytest.mark.parametrize("cin,feat,res", [(10, 32, 2), (16, 16, 1), (12, 24, 1)])
def test_generator_shapes_the_split_layout_cannot_carry_run_in_exact_fp32(cin, feat, res):
    """the reference accepts any in_channels; more than 8 input channels (outside the split-fp16 input layout) are served by the exact-fp32 MFMA kernels instead
    of being refused ('refuses shapes the reference accepts') — same output tolerance, also through forward_pair.  init_features % 8 != 0 stays a
    loud refusal: tools/gen_shape_probe.py measured 12 / 20 features WRONG in the exact-fp32 mode that round 3 accepted them in."""
    from smirk_amd import SmirkGenerator
    sd = G.synth_state_dict(in_channels=cin, out_channels=3, features=feat, res_blocks=res, seed=77)
    m = SmirkGenerator(in_channels=cin, out_channels=3, init_features=feat, res_blocks=res)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(cin * 100 + feat)
    x = torch.rand(2, cin, 32, 48, generator=g)
    y = G.forward(sd, x, res_blocks=res)
    with torch.no_grad(), pytest.warns(UserWarning, match="exact-fp32"):
        out = m(x.cuda())
    assert (out.cpu() - y).abs().max().item() < OUT_TOL
    with torch.no_grad():
        k = cin // 2
        out2 = m.forward_pair(x[:, :k].contiguous().cuda(), x[:, k:].contiguous().cuda())
    assert torch.equal(out2, out)
    with pytest.raises(Exception, match="multiple of 16"):                    # the reference's torch.cat fails on such sizes too
        m(torch.rand(1, cin, 40, 40).cuda())
    with pytest.raises(Exception, match="init_features % 8"):
        SmirkGenerator(in_channels=6, out_channels=3, init_features=12, res_blocks=1)


def test_generator_in_channels_above_8_refuses_gradients_up_front():
    """in_channels > 8 is an inference-only shape (exact-fp32 kernels): train mode and an eval-mode input gradient say so when they are ASKED for, not when a
    backward pass finally needs them."""
    from smirk_amd import SmirkGenerator, SmirkHipError
    m = SmirkGenerator(in_channels=10, out_channels=3, init_features=16, res_blocks=1).cuda().eval()
    x = torch.rand(1, 10, 32, 32, device="cuda", requires_grad=True)
    with pytest.raises(SmirkHipError, match="gradient with respect to the input"):
        m(x)
    m.train()
    with pytest.raises(SmirkHipError, match="in_channels <= 8"):
        m(x.detach())


@pytest.mark.parametrize("chains", ["0", "2", "3"])
@pytest.mark.parametrize("B", [3, 20])
def test_generator_sub_batch_chains_are_bitwise_the_single_chain(chains, B, monkeypatch):
    """smirk_generator_forward runs the H/8 + H/16 section as 1 / 2 / 3 sub-batch chains on side streams (csrc/network.hip); every frame's result must be the
    single-chain result bit for bit, for batches that do not divide evenly too.  $SMIRK_GEN_SPLIT_CHAINS forces the chain count (read per call)."""
    from smirk_amd import SmirkGenerator
    sd = G.synth_state_dict()
    m = SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = A.synth_generator_input(B, seed=31).cuda()
    monkeypatch.setenv("SMIRK_GEN_SPLIT_CHAINS", "0")
    with torch.no_grad():
        ref = m(x).clone()
    monkeypatch.setenv("SMIRK_GEN_SPLIT_CHAINS", chains)
    with torch.no_grad():
        for _ in range(3):                                                       # repeated: the side streams and events are reused across calls
            y = m(x)
            torch.cuda.synchronize()
            assert torch.equal(y, ref)
