"""BASELINE config 1 ("demo.py single image, plumbing"): the REFERENCE's own demo.py, unmodified, executed on top of the drop-in classes through
the `integration/shim` package, in the build container (skipped where /root/reference does not exist, e.g. the GPU box).
Everything up to the first device call runs: imports, the three constructors with the reference's arguments, the strict `load_state_dict`
of a `smirk_encoder.` / `smirk_generator.`-prefixed checkpoint (demo.py:54-66), `.eval()`, `FLAME().to(...)`, `Renderer().to(...)`, the
pre-processing up to `smirk_encoder(cropped_image)` — which must then fail LOUDLY, because the HIP path has no CPU fallback."""
import os
import runpy
import sys
import traceback
import types

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "demo.py")), reason="the reference checkout exists in the build container only")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def test_reference_demo_runs_on_the_shim_until_the_first_device_call(in_sandbox, monkeypatch, tmp_path):
    from oracle import generator_ref as G, mobilenet_ref as M
    import smirk_amd
    # synthetic checkpoint in the reference's wire format (base_trainer.py:222-231): one flat dict, module-prefixed keys
    ckpt = {"smirk_encoder." + k: v for k, v in M.synth_encoder_state_dict().items()}
    ckpt.update({"smirk_generator." + k: v for k, v in G.synth_state_dict(calibrate=False).items()})
    ckpt_path = str(tmp_path / "SMIRK_em1.pt")
    torch.save(ckpt, ckpt_path)

    calls = []
    rng = np.random.default_rng(0)
    img = (rng.uniform(0, 255, (224, 224, 3))).astype(np.uint8)
    kpt = np.stack([rng.uniform(60, 160, 478), rng.uniform(60, 160, 478), np.zeros(478)], 1)
    cv2 = _stub("cv2", imread=lambda p: img.copy(), cvtColor=lambda a, code: a[..., ::-1].copy(), COLOR_BGR2RGB=4, COLOR_RGB2BGR=4,
                resize=lambda a, size: a, imwrite=lambda *a: calls.append("imwrite"))
    sk = _stub("skimage"); skt = _stub("skimage.transform", estimate_transform=lambda *a, **k: None, warp=lambda *a, **k: None)
    mp_utils = _stub("utils.mediapipe_utils", run_mediapipe=lambda image: kpt)
    utils_pkg = _stub("utils"); utils_pkg.__path__ = []
    ds_pkg = _stub("datasets"); ds_pkg.__path__ = []
    base_ds = _stub("datasets.base_dataset", create_mask=lambda lm, shape: np.ones(shape, np.float32))
    for name, mod in (("cv2", cv2), ("skimage", sk), ("skimage.transform", skt), ("utils", utils_pkg), ("utils.mediapipe_utils", mp_utils),
                      ("datasets", ds_pkg), ("datasets.base_dataset", base_ds)):
        monkeypatch.setitem(sys.modules, name, mod)
    for name in [n for n in sys.modules if n == "src" or n.startswith("src.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(os.path.join(REPO, "integration", "shim"))
    monkeypatch.setattr(sys, "argv", ["demo.py", "--input_path", "x.png", "--device", "cpu", "--checkpoint", ckpt_path, "--use_smirk_generator",
                                      "--out_path", str(tmp_path / "out")])
    with pytest.raises(smirk_amd.SmirkHipError) as ei:
        runpy.run_path(os.path.join(REF, "demo.py"), run_name="__main__")
    assert "no CPU fallback" in str(ei.value)
    frames = [f for f in traceback.extract_tb(ei.tb) if f.filename == os.path.join(REF, "demo.py")]
    assert frames and "smirk_encoder(cropped_image)" in (frames[-1].line or ""), "demo.py must have reached its first forward call"


def test_shim_exposes_the_reference_module_paths(monkeypatch):
    """every `from src... import ...` of demo.py:5-10, demo_video.py and smirk_trainer.py:4-8 resolves on the shim"""
    import importlib
    for name in [n for n in sys.modules if n == "src" or n.startswith("src.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(os.path.join(REPO, "integration", "shim"))
    for mod, names in (("src.smirk_encoder", ["SmirkEncoder"]), ("src.FLAME.FLAME", ["FLAME"]), ("src.renderer.renderer", ["Renderer"]),
                       ("src.smirk_generator", ["SmirkGenerator"]),
                       ("src.utils.masking", ["load_probabilities_per_FLAME_triangle", "mesh_based_mask_uniform_faces", "masking", "transfer_pixels",
                                              "point2ind", "triangle_area", "random_barycentric"])):
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
        assert m.__file__.startswith(os.path.join(REPO, "integration", "shim"))


def test_graph_cycle_modules_refuses_eval_mode():
    """smirk_amd.cycle.graph_cycle_modules captures the TRAIN-mode path only (batch-statistics BatchNorm + backward); asking for it in eval mode is an error, raised
    before anything touches a device"""
    import torch
    from smirk_amd.cycle import CYCLE_GRAD_KEYS, _CycleEncoder, graph_cycle_modules

    class Tiny(torch.nn.Module):
        def forward(self, x):
            return {k: x.mean() * (i + 1) for i, k in enumerate(CYCLE_GRAD_KEYS + ("pose_params", "cam"))}

    g, e = torch.nn.Conv2d(6, 3, 1).eval(), Tiny().eval()
    with pytest.raises(ValueError):
        graph_cycle_modules(g, e, torch.zeros(1, 6, 16, 16), torch.zeros(1, 3, 16, 16))
    out = _CycleEncoder(Tiny())(torch.ones(1, 3, 4, 4, requires_grad=True))          # the wrapper detaches what the cycle loss does not differentiate
    assert all(out[k].requires_grad for k in CYCLE_GRAD_KEYS) and not out["pose_params"].requires_grad and not out["cam"].requires_grad
