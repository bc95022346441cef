"""The compute of the reference trainer's cycle path (BASELINE config 5; smirk_trainer.py:184-332 `step2`, :349-382 `step`) over the smirk_amd modules.

The trainer itself — parameter augmentation, dataset, logging, optimiser schedule — is the reference's and stays the reference's (caller of
the hot path, out of scope).  What it asks of the four modules per step is restated here so that tests and bench.py can drive exactly that sequence:

    with no_grad:   FLAME(encoder_output), Renderer(...)                       smirk_trainer.py:247-249   (sampling of the source points)
                    FLAME(augmented params), Renderer(...)  -> rendered        smirk_trainer.py:252-254
                    mesh_based_mask_uniform_faces x2, transfer_pixels, mask    smirk_trainer.py:262-289
    masking(...)    -> masked image                                            smirk_trainer.py:290-291
    generator(cat[rendered, masked])          TRAIN mode (batch-statistic BN)  smirk_trainer.py:293
    encoder(reconstructed)                    TRAIN mode                       smirk_trainer.py:297
    FLAME / Renderer of the re-encoded parameters (visualisation only)         smirk_trainer.py:299-300
    cycle loss = mse(exp) + 10 mse(jaw) + 10 mse(eyelid) [+ mse(shape)]        smirk_trainer.py:304-313
    backward; clip_grad_norm_(generator, 0.1); optimiser steps                 smirk_trainer.py:367-376

Multi-GPU (SURVEY.md §8(e) "C2"): batch-sharded data parallel, gradients averaged with ONE bucketed RCCL all-reduce per dtype after backward
(`allreduce_gradients`); xGMI is point-to-point, so few large buckets (default 64 MiB: the generator's 125 MB of fp32 gradients go out as two) rather than
per-parameter calls.  BatchNorm statistics stay per rank, like torch DDP without SyncBatchNorm (the reference has no DDP wrapper at all).
"""
import torch
import torch.nn.functional as F


def cycle_loss(recon_feats, flame_feats, use_eyelids=True, generator_frozen=False):
    """smirk_trainer.py:304-313"""
    loss = 1.0 * F.mse_loss(recon_feats['expression_params'], flame_feats['expression_params']) + \
        10.0 * F.mse_loss(recon_feats['jaw_params'], flame_feats['jaw_params'])
    if use_eyelids:
        loss = loss + 10.0 * F.mse_loss(recon_feats['eyelid_params'], flame_feats['eyelid_params'])
    if not generator_frozen:
        loss = loss + 1.0 * F.mse_loss(recon_feats['shape_params'], flame_feats['shape_params'])
    return loss


def render_second_path(flame, renderer, encoder_output, flame_feats, img, masks, face_probabilities, masking_utils, mask_ratio=0.01,
                       mask_dilation_radius=10, Ke=1):
    """smirk_trainer.py:245-291: everything between the augmented parameters and the generator's input.  Returns (rendered, masked)."""
    with torch.no_grad():
        flame_output = flame.forward(encoder_output)
        rendered_output = renderer.forward(flame_output['vertices'], encoder_output['cam'])
        flame_output.update(rendered_output)
        flame_output_2nd = flame.forward(flame_feats)
        cam = encoder_output['cam'] if Ke == 1 else encoder_output['cam'].repeat(Ke, 1)
        renderer_output_2nd = renderer.forward(flame_output_2nd['vertices'], cam)
        rendered_2nd = renderer_output_2nd['rendered_img'].detach()
        points1, coords = masking_utils.mesh_based_mask_uniform_faces(flame_output['transformed_vertices'], flame_faces=flame.faces_tensor,
                                                                      face_probabilities=face_probabilities, mask_ratio=mask_ratio)
        coords['sampled_faces_indices'] = coords['sampled_faces_indices'].repeat(Ke, 1)
        coords['barycentric_coords'] = coords['barycentric_coords'].repeat(Ke, 1, 1)
        points2, coords = masking_utils.mesh_based_mask_uniform_faces(renderer_output_2nd['transformed_vertices'], flame_faces=flame.faces_tensor,
                                                                      face_probabilities=face_probabilities, mask_ratio=mask_ratio, coords=coords)
        extra_points = masking_utils.transfer_pixels(img.repeat(Ke, 1, 1, 1), points1.repeat(Ke, 1, 1), points2)
        rendered_mask = (rendered_2nd > 0).all(dim=1, keepdim=True).float()
    masked_2nd = masking_utils.masking(img.repeat(Ke, 1, 1, 1), masks.repeat(Ke, 1, 1, 1), extra_points, mask_dilation_radius, rendered_mask=rendered_mask,
                                       extra_noise=True, random_mask=0.005)
    return rendered_2nd, masked_2nd


def cycle_forward(generator, encoder, rendered, masked, flame_feats, freeze_generator=False, use_eyelids=True):
    """smirk_trainer.py:293-313: generator -> encoder -> cycle loss.  Returns (loss, reconstructed image, re-encoded parameters)."""
    recon = generator(torch.cat([rendered, masked], dim=1).detach())
    if freeze_generator:
        recon = recon.detach()
    feats = encoder(recon)
    return cycle_loss(feats, flame_feats, use_eyelids, freeze_generator), recon, feats


CYCLE_GRAD_KEYS = ("expression_params", "jaw_params", "eyelid_params", "shape_params")        # what cycle_loss differentiates (smirk_trainer.py:304-313)


class _CycleEncoder(torch.nn.Module):
    """SmirkEncoder as the cycle path uses it: the parameters the cycle loss does not touch (pose, camera — smirk_trainer.py:304-313 has no term for them)
    leave detached, so a captured backward graph contains no backward pass of the (frozen) pose backbone — exactly what eager autograd prunes at run time."""

    def __init__(self, encoder, grad_keys=CYCLE_GRAD_KEYS):
        super().__init__()
        self.encoder, self.grad_keys = encoder, tuple(grad_keys)

    def forward(self, img):
        out = self.encoder(img)
        return {k: (v if k in self.grad_keys else v.detach()) for k, v in out.items()}


class _CycleGenerator(torch.nn.Module):
    """pass-through wrapper: make_graphed_callables patches the `forward` of the module it is given — the user's SmirkGenerator stays as it is"""

    def __init__(self, generator):
        super().__init__()
        self.generator = generator

    def forward(self, x):
        return self.generator(x)


def graph_cycle_modules(generator, encoder, generator_input, encoder_input, grad_keys=CYCLE_GRAD_KEYS, warmup=3):
    """TRAIN-mode forward AND backward of the two CNNs captured into HIP graphs (`torch.cuda.make_graphed_callables`): the cycle step launches ~1300
    kernels whose Python/ctypes enqueue (~40 ms) is as long as their execution on one MI355X; replaying four graphs (generator / encoder, forward /
    backward) costs the host well under a millisecond each.  Everything the two `torch.autograd.Function`s do is stream-ordered device work on the
    current stream (kernel launches of libsmirk_hip.so, allocator calls, no host synchronisation), which is what makes them capturable; BatchNorm running
    statistics and `num_batches_tracked` are updated by captured kernels, parameters are read in place, so optimiser steps between replays are seen.
    The sample tensors fix the shapes: `generator_input` [B, 6, H, W] (no gradient needed, as in smirk_trainer.py:293), `encoder_input` [B, 3, H, W].
    Returns (generator_for_cycle, encoder_for_cycle): wrappers to call exactly like the modules (same shapes as the samples); the modules themselves are
    left untouched and keep launching kernel by kernel.
    Note: capturing runs the modules `warmup` + 1 times, which advances BatchNorm running statistics like that many training steps.
    Note: replays write running_mean / running_var / num_batches_tracked from captured kernels, so torch's `_version` counters do not move; the modules'
    eval-mode weight caches (folded BatchNorm) are therefore dropped on every train() <-> eval() transition (SmirkGenerator.train,
    MobileNetV3Features.train) — call `.eval()` AFTER the last replay, as `nn.Module` users do anyway."""
    if not (generator.training and encoder.training):
        raise ValueError("graph_cycle_modules captures the TRAIN-mode path: call .train() first")
    enc = _CycleEncoder(encoder, grad_keys)
    gen = _CycleGenerator(generator)
    enc.train(); gen.train()
    # The samples only fix shapes, dtype and device: make_graphed_callables runs its warm-up and capture passes ON them (they become the static input buffers that
    # every replay overwrites).  Constant samples (callers pass zeros) make every BatchNorm's batch variance exactly 0, i.e. invstd = 1 / sqrt(eps) = 316, and the
    # warm-up BACKWARD multiplies the gradient by that factor in each of the generator's 28 BatchNorm layers: inf / NaN gradients that the split-fp16 range flag
    # (include/smirk_hip.h smirk_range_flag_peek) rightly reports at the next forward.  Warm up on uniform noise instead (seeded: the capture is reproducible).
    g = torch.Generator(device=generator_input.device).manual_seed(20240)
    gi = torch.rand(generator_input.shape, generator=g, device=generator_input.device, dtype=generator_input.dtype)
    ei = torch.rand(encoder_input.shape, generator=g, device=encoder_input.device, dtype=encoder_input.dtype).requires_grad_(True)
    # Capture on ONE stream: eager training runs the three backbones on three side streams, but a capture that forks to pre-existing side streams and joins
    # them back (with one backbone's backward pruned) made hipStreamEndCapture segfault in some test orders on ROCm 7 (round 3: deterministic after
    # tests/test_conv_gpu.py, never when the test ran alone).  The replayed graph is a chain either way — replay was measured SLOWER than eager launches
    # (DESIGN.md 8.9), this path exists for its host-time saving — so nothing is lost by recording the reference order.
    import gc
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    if warmup < 1:
        raise ValueError("graph_cycle_modules needs warmup >= 1: the first eager pass seals the weight-packing plans (a pageable host-to-device copy, illegal inside a capture)")
    # the one-stream order is requested from THIS encoder object only (an attribute its forward reads), not through the process environment: other threads
    # or modules running an encoder meanwhile keep their three streams
    prev = getattr(encoder, "_train_serial", False)
    encoder._train_serial = True
    try:
        return torch.cuda.make_graphed_callables((gen, enc), ((gi,), (ei,)), num_warmup_iters=warmup, allow_unused_input=True)
    finally:
        encoder._train_serial = prev


def set_train_arith(generator, encoder, arith):
    """Arithmetic of the convolutions (forward, data gradient, weight gradient) of the two CNNs in TRAIN mode: "f16x3" (fp32-class, default) or "f16x1" (one
    fp16 MFMA per product block: the 16-bit class BASELINE config 5 names; generator_train.TRAIN_ARITH).  Everything else — BatchNorm, FLAME, renderer,
    losses, Adam — is unchanged."""
    from .generator_train import TRAIN_ARITH
    if arith not in TRAIN_ARITH:
        raise ValueError(f"train_arith must be one of {TRAIN_ARITH}")
    if generator is not None:
        generator.train_arith = arith
    if encoder is not None:
        for name in ("pose_encoder", "shape_encoder", "expression_encoder"):
            getattr(encoder, name).encoder.train_arith = arith


def allreduce_gradients(params, group=None, bucket_bytes=64 << 20, average=True, force_collective=False):
    """Data-parallel gradient exchange (C2): flatten the gradients into few large buckets, one all-reduce each (RCCL over xGMI on GPUs, gloo in the
    CPU tests), scatter back.  Parameters without a gradient on this rank are skipped on EVERY rank only if they are skipped everywhere — the trainer
    freezes the same modules on all ranks, and the bucket layout is derived from `requires_grad`, not from `.grad is None`, so ranks cannot disagree."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return 0
    if dist.get_world_size(group) == 1 and not force_collective:       # force_collective: a one-GPU job still drives RCCL (tests/test_rccl_gpu.py)
        return 0
    world = dist.get_world_size(group)
    todo = [p for p in params if p.requires_grad]
    buckets, cur, size = [], [], 0
    for p in todo:
        n = p.numel() * p.element_size()
        if cur and (size + n > bucket_bytes or p.dtype != cur[0].dtype):
            buckets.append(cur); cur, size = [], 0
        cur.append(p); size += n
    if cur:
        buckets.append(cur)
    handles = []
    for b in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
        handles.append((b, flat, dist.all_reduce(flat, group=group, async_op=True)))
    for b, flat, h in handles:
        h.wait()
        if average:
            flat.div_(world)
        o = 0
        for p in b:
            n = p.numel()
            g = flat[o:o + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            o += n
    return len(buckets)
