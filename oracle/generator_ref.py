"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torch-CPU fp32 functional restatement of src/smirk_generator.py (SmirkGenerator.forward :51-86,
_block :88-119, ResnetBlock :121-178) driven by a reference-keyed state_dict, eval-mode BatchNorm.
Floating-point kernel => a torch fp32 reference is the oracle (tolerances are stated in the tests).
Pinned against the reference class itself: oracle/make_golden.py runs it (build container) -> tests/golden/generator_golden.npz,
compared with this restatement by tests/test_cpu_suite.py::test_generator_oracle_vs_reference_golden.
"""
import torch
import torch.nn.functional as F


def synth_state_dict(in_channels=6, out_channels=3, features=32, res_blocks=5, seed=1234, calibrate=True):
    """Seeded synthetic weights with exactly the reference's 178 keys (SURVEY.md §8(b),(d)).  With `calibrate` the BN
    running stats are set to (perturbed) batch statistics of 4 synthetic inputs and the last conv is scaled to unit-std
    logits, so activations stay O(1) and the sigmoid is not saturated (a saturated output would hide kernel errors)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k, bias=False):
        fan_out = co * k * k
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * (2.0 / fan_out) ** 0.5
        if bias:
            sd[name + ".bias"] = torch.randn(co, generator=g) * 0.05

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=g) + 0.5
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_var"] = torch.rand(c, generator=g) + 0.5
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def block(mod, nm, ci, co):
        conv(f"{mod}.{nm}conv1", co, ci, 3); bn(f"{mod}.{nm}norm1", co)
        conv(f"{mod}.{nm}conv2", co, co, 3); bn(f"{mod}.{nm}norm2", co)

    f = features
    block("encoder1", "enc1", in_channels, f); block("encoder2", "enc2", f, 2 * f)
    block("encoder3", "enc3", 2 * f, 4 * f); block("encoder4", "enc4", 4 * f, 8 * f)
    block("bottleneck", "bottleneck", 8 * f, 16 * f)
    for k in range(res_blocks):
        p = f"resnet_blocks.{k}.conv_block"
        conv(p + ".1", 16 * f, 16 * f, 3); bn(p + ".2", 16 * f)
        conv(p + ".5", 16 * f, 16 * f, 3); bn(p + ".6", 16 * f)
    for lvl, (ci, co) in zip((4, 3, 2, 1), ((16 * f, 8 * f), (8 * f, 4 * f), (4 * f, 2 * f), (2 * f, f))):
        # ConvTranspose2d weight is [Cin, Cout, 2, 2]
        sd[f"upconv{lvl}.weight"] = torch.randn(ci, co, 2, 2, generator=g) * (1.0 / ci) ** 0.5
        sd[f"upconv{lvl}.bias"] = torch.randn(co, generator=g) * 0.05
        block(f"decoder{lvl}", f"dec{lvl}", 2 * co, co)
    conv("conv", out_channels, f, 1, bias=True)
    if calibrate:
        from .assets import synth_generator_input
        if in_channels == 6:
            xc = synth_generator_input(4, seed=777)
        else:                                                        # other widths (tests of non-default shapes): a small seeded random batch
            xc = torch.rand(2, in_channels, 64, 64, generator=torch.Generator().manual_seed(777))
        forward(sd, xc, res_blocks, _calib=g)
    return sd


_CALIB = None


def _bn(x, sd, p):
    if _CALIB is not None:            # calibration pass of synth_state_dict: write perturbed batch stats into sd
        c = x.shape[1]
        mu, var = x.mean((0, 2, 3)), x.var((0, 2, 3), unbiased=False)
        sd[p + ".running_mean"] = mu + torch.randn(c, generator=_CALIB) * 0.1 * var.sqrt()
        sd[p + ".running_var"] = var * (torch.rand(c, generator=_CALIB) + 0.5)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=1e-5)


def _block(x, sd, mod, nm):
    x = F.relu(_bn(F.conv2d(x, sd[f"{mod}.{nm}conv1.weight"], padding=1), sd, f"{mod}.{nm}norm1"))
    x = F.relu(_bn(F.conv2d(x, sd[f"{mod}.{nm}conv2.weight"], padding=1), sd, f"{mod}.{nm}norm2"))
    return x


def forward(sd, x, res_blocks=5, taps=None, _calib=None, grad=False):
    """x [B,6,224,224] fp32 -> [B,3,224,224]; optional `taps` dict collects intermediates for layer-wise parity.
    grad=True keeps the autograd graph (eval-mode BatchNorm inside a differentiated graph, smirk_trainer.py:108-113: dL/dx of the frozen generator)."""
    global _CALIB
    t = taps if taps is not None else {}
    _CALIB = _calib
    try:
        return _forward(sd, x, res_blocks, t, grad)
    finally:
        _CALIB = None


def _forward(sd, x, res_blocks, t, grad=False):
    with torch.set_grad_enabled(grad):
        e1 = _block(x, sd, "encoder1", "enc1"); t["enc1"] = e1
        e2 = _block(F.max_pool2d(e1, 2, 2), sd, "encoder2", "enc2"); t["enc2"] = e2
        e3 = _block(F.max_pool2d(e2, 2, 2), sd, "encoder3", "enc3"); t["enc3"] = e3
        e4 = _block(F.max_pool2d(e3, 2, 2), sd, "encoder4", "enc4"); t["enc4"] = e4
        b = _block(F.max_pool2d(e4, 2, 2), sd, "bottleneck", "bottleneck"); t["bottleneck"] = b
        for k in range(res_blocks):
            p = f"resnet_blocks.{k}.conv_block"
            y = F.conv2d(F.pad(b, (1, 1, 1, 1), mode="reflect"), sd[p + ".1.weight"])
            y = F.relu(_bn(y, sd, p + ".2"))
            y = F.conv2d(F.pad(y, (1, 1, 1, 1), mode="reflect"), sd[p + ".5.weight"])
            b = b + _bn(y, sd, p + ".6")
        t["res"] = b
        d = b
        for lvl, skip in ((4, e4), (3, e3), (2, e2), (1, e1)):
            d = F.conv_transpose2d(d, sd[f"upconv{lvl}.weight"], sd[f"upconv{lvl}.bias"], stride=2)
            d = _block(torch.cat((d, skip), 1), sd, f"decoder{lvl}", f"dec{lvl}")
            t[f"dec{lvl}"] = d
        if _CALIB is not None:
            lg = F.conv2d(d, sd["conv.weight"])
            sd["conv.weight"] = sd["conv.weight"] / lg.std().clamp_min(1e-6)
            sd["conv.bias"] = -(lg / lg.std().clamp_min(1e-6)).mean((0, 2, 3)) + sd["conv.bias"]
        t["logits"] = F.conv2d(d, sd["conv.weight"], sd["conv.bias"])
        return torch.sigmoid(t["logits"])


# ------------------------------------------------------------------------------------------------------------------------------------
# TRAIN mode (BASELINE config 5 / VERDICT r1 row J1): batch-statistics BatchNorm + autograd, restated functionally.
# Follows what nn.BatchNorm2d.train() does inside the reference's SmirkGenerator (smirk_generator.py:88-119 `_block`, :121-178 ResnetBlock):
# normalise with the batch mean / biased variance, update running_mean / running_var with momentum 0.1 (running_var takes the UNBIASED batch
# variance), bump num_batches_tracked.  Pinned against the real reference class by oracle/make_train_golden.py (outputs, input gradient and
# every parameter gradient identical) and tests/golden/generator_train_golden.npz.
# ------------------------------------------------------------------------------------------------------------------------------------
def train_forward(params, buffers, x, res_blocks=5, momentum=0.1, eps=1e-5):
    """params: {name: tensor (requires_grad as the caller wishes)}, buffers: {name: tensor} (running stats; UPDATED IN PLACE like the module does).
    x [B,6,H,W] -> sigmoid image, differentiable w.r.t. x and params."""
    def bn(t, p):
        return F.batch_norm(t, buffers[p + ".running_mean"], buffers[p + ".running_var"], params[p + ".weight"], params[p + ".bias"],
                            training=True, momentum=momentum, eps=eps)

    def block(t, mod, nm):
        t = F.relu(bn(F.conv2d(t, params[f"{mod}.{nm}conv1.weight"], padding=1), f"{mod}.{nm}norm1"))
        return F.relu(bn(F.conv2d(t, params[f"{mod}.{nm}conv2.weight"], padding=1), f"{mod}.{nm}norm2"))

    e1 = block(x, "encoder1", "enc1")
    e2 = block(F.max_pool2d(e1, 2, 2), "encoder2", "enc2")
    e3 = block(F.max_pool2d(e2, 2, 2), "encoder3", "enc3")
    e4 = block(F.max_pool2d(e3, 2, 2), "encoder4", "enc4")
    b = block(F.max_pool2d(e4, 2, 2), "bottleneck", "bottleneck")
    for k in range(res_blocks):
        p = f"resnet_blocks.{k}.conv_block"
        y = F.relu(bn(F.conv2d(F.pad(b, (1, 1, 1, 1), mode="reflect"), params[p + ".1.weight"]), p + ".2"))
        b = b + bn(F.conv2d(F.pad(y, (1, 1, 1, 1), mode="reflect"), params[p + ".5.weight"]), p + ".6")
    d = b
    for lvl, skip in ((4, e4), (3, e3), (2, e2), (1, e1)):
        d = F.conv_transpose2d(d, params[f"upconv{lvl}.weight"], params[f"upconv{lvl}.bias"], stride=2)
        d = block(torch.cat((d, skip), 1), f"decoder{lvl}", f"dec{lvl}")
    return torch.sigmoid(F.conv2d(d, params["conv.weight"], params["conv.bias"]))


def split_state_dict(sd, dtype=torch.float32):
    """state_dict -> (params that take gradients, buffers) as fresh leaf tensors (dtype=torch.float64 gives the high-precision arbiter)"""
    params = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()
              if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))}
    buffers = {k: v.clone().to(dtype) for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
    return params, buffers


def train_step(sd, x, loss_weights, res_blocks=5, dtype=torch.float32):
    """One forward + backward of loss = sum(y * loss_weights) in train mode.  Returns y, loss, dL/dx, {param: grad}, {buffer: updated running stat}."""
    params, buffers = split_state_dict(sd, dtype)
    x = x.clone().to(dtype).requires_grad_(True)
    y = train_forward(params, buffers, x, res_blocks)
    loss = (y * loss_weights.to(dtype)).sum()
    loss.backward()
    return y.detach(), float(loss.detach()), x.grad, {k: v.grad for k, v in params.items()}, buffers
