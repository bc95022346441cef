"""Seeded synthetic assets and inputs for benchmarks, smoke runs and tests (no licence-gated data, no network).  Bench / test helper at the repo root, next to
bench.py: NOT part of the importable product (smirk_amd/ reads nothing under tests/).

The reference reads its assets by cwd-relative path (FLAME.py:50-51,54,81-82,94,111; renderer.py:50,54,65).  The real
FLAME2020/generic_model.pkl is licence-gated, so `write_sandbox` builds an ``assets/`` tree in the on-disk formats the loaders
expect: the public topology / landmark embeddings / eyelid blendshapes / region masks packed in
``tests/golden/assets_bundle.npz`` plus a seeded synthetic FLAME model.  Both the GPU path and the CPU oracle
consume exactly these bits, generated on the CPU.
"""
import os
import pickle

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
BUNDLE = os.path.join(REPO, "tests", "golden", "assets_bundle.npz")
V = 5023
F = 9976


def load_bundle(path=BUNDLE):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def synth_flame_model(bundle, seed=2020, basis="random"):
    """Seeded synthetic stand-in for FLAME2020/generic_model.pkl.

    Keys / shapes are those FLAME.__init__ reads (FLAME.py:54-78): plain ndarrays so no
    chumpy is needed to unpickle.

    basis="random" (the SURVEY.md §8(d) definition, used by every parity test): i.i.d. Gaussian blendshape directions — neighbouring vertices move
    independently, so deformed triangles are stretched to ~21 x 22-pixel boxes at 224 x 224.
    basis="smooth": every direction is a low-frequency vector field over the template (a random axis times a sinusoid of <= 2 cycles across the head)
    with the same per-direction RMS — neighbouring vertices move together like a real statistical shape model, triangles keep their ~4-pixel size.
    Only the rasteriser's work depends on this (bench.py --flame-basis smooth measures its share on realistic triangles).
    """
    rng = np.random.default_rng(seed)
    vt = bundle["obj_verts"].astype(np.float64)
    vt = vt - vt.mean(0, keepdims=True)
    decay = 0.985 ** np.arange(400)
    if basis == "smooth":
        ext = np.abs(vt).max()
        def fields(n, amp):
            axis = rng.standard_normal((n, 3)); axis /= np.linalg.norm(axis, axis=1, keepdims=True)
            freq = rng.standard_normal((n, 3)); freq *= (rng.uniform(0.3, 2.0, (n, 1)) / np.linalg.norm(freq, axis=1, keepdims=True))
            phase = rng.uniform(0, 2 * np.pi, n)
            wave = np.sin(2 * np.pi * (vt / (2 * ext)) @ freq.T + phase[None, :])          # [V, n]
            wave /= np.sqrt((wave ** 2).mean(0, keepdims=True))                            # unit RMS per direction
            return wave[:, None, :] * axis.T[None, :, :] * np.sqrt(3.0) * amp               # [V, 3, n], per-component RMS ~ amp
        shapedirs = fields(400, 5e-3 * decay[None, None, :])
        posedirs = fields(36, 1e-3)
    elif basis == "random":
        shapedirs = rng.standard_normal((V, 3, 400)) * 5e-3 * decay[None, None, :]
        posedirs = rng.standard_normal((V, 3, 36)) * 1e-3
    else:
        raise ValueError("basis must be 'random' or 'smooth'")
    jr = np.abs(rng.standard_normal((5, V)))
    jr /= jr.sum(1, keepdims=True)
    w = np.abs(rng.standard_normal((V, 5)))
    w /= w.sum(1, keepdims=True)
    kintree = np.array([[2 ** 32 - 1, 0, 1, 1, 1], [0, 1, 2, 3, 4]], dtype=np.int64)
    return dict(f=bundle["obj_faces"].astype(np.uint32), v_template=vt, shapedirs=shapedirs,
                posedirs=posedirs, J_regressor=jr, kintree_table=kintree, weights=w)


def write_obj(path, verts, uvs, faces, tfaces):
    with open(path, "w") as fh:
        fh.write("# synthetic-sandbox copy of the head template topology\n")
        for p in verts:
            fh.write("v %.6f %.6f %.6f\n" % (p[0], p[1], p[2]))
        for t in uvs:
            fh.write("vt %.6f %.6f\n" % (t[0], t[1]))
        for a, b in zip(faces, tfaces):
            fh.write("f %d/%d %d/%d %d/%d\n" % (a[0] + 1, b[0] + 1, a[1] + 1, b[1] + 1, a[2] + 1, b[2] + 1))


def write_sandbox(root, bundle=None, seed=2020, basis="random"):
    """Create ``root/assets/...`` with every file FLAME() and Renderer() open. Returns root.  basis: see synth_flame_model."""
    import torch
    bundle = bundle if bundle is not None else load_bundle()
    a = os.path.join(root, "assets")
    os.makedirs(os.path.join(a, "FLAME2020"), exist_ok=True)
    os.makedirs(os.path.join(a, "FLAME_masks"), exist_ok=True)
    os.makedirs(os.path.join(a, "mediapipe_landmark_embedding"), exist_ok=True)
    pkl = os.path.join(a, "FLAME2020", "generic_model.pkl")
    if not os.path.exists(pkl):
        with open(pkl, "wb") as fh:
            pickle.dump(synth_flame_model(bundle, seed, basis), fh, protocol=2)
    np.save(os.path.join(a, "l_eyelid.npy"), bundle["l_eyelid"])
    np.save(os.path.join(a, "r_eyelid.npy"), bundle["r_eyelid"])
    emb = dict(
        static_lmk_faces_idx=bundle["static_lmk_faces_idx"],
        static_lmk_bary_coords=bundle["static_lmk_bary_coords"],
        dynamic_lmk_faces_idx=torch.from_numpy(bundle["dynamic_lmk_faces_idx"]),
        dynamic_lmk_bary_coords=torch.from_numpy(bundle["dynamic_lmk_bary_coords"]),
        full_lmk_faces_idx=bundle["full_lmk_faces_idx"],
        full_lmk_bary_coords=bundle["full_lmk_bary_coords"],
    )
    np.save(os.path.join(a, "landmark_embedding.npy"), np.array(emb, dtype=object), allow_pickle=True)
    write_obj(os.path.join(a, "head_template.obj"), bundle["obj_verts"], bundle["obj_uvs"],
              bundle["obj_faces"], bundle["obj_tfaces"])
    masks = {k[5:]: bundle[k] for k in bundle if k.startswith("mask_")}
    with open(os.path.join(a, "FLAME_masks", "FLAME_masks.pkl"), "wb") as fh:
        pickle.dump(masks, fh, protocol=2)
    tris = {k[4:]: bundle[k].astype(np.int64) for k in bundle if k.startswith("tri_")}
    if tris:
        np.save(os.path.join(a, "FLAME_masks", "FLAME_masks_triangles.npy"), np.array(tris, dtype=object), allow_pickle=True)
    np.savez(os.path.join(a, "mediapipe_landmark_embedding", "mediapipe_landmark_embedding.npz"),
             lmk_face_idx=bundle["mp_lmk_face_idx"], lmk_b_coords=bundle["mp_lmk_b_coords"],
             landmark_indices=bundle["mp_landmark_indices"])
    return root


# ---------------------------------------------------------------------------------------
# seeded synthetic inputs; generated on CPU so oracle and GPU see the same bits
# ---------------------------------------------------------------------------------------
def synth_flame_params(B, seed=0, n_shape=300, n_exp=50):
    """Config-2 style FLAME parameters (shape/exp/pose/jaw/eyelid), float32 numpy."""
    rng = np.random.default_rng(seed)
    p = dict(
        shape_params=rng.standard_normal((B, n_shape)),
        expression_params=np.clip(rng.standard_normal((B, n_exp)) * 1.5, -4, 4),
        pose_params=rng.uniform(-0.4, 0.4, (B, 3)),
        jaw_params=np.stack([rng.uniform(0, 0.5, B), rng.uniform(-.2, .2, B), rng.uniform(-.2, .2, B)], 1),
        eyelid_params=rng.uniform(0, 1, (B, 2)),
    )
    return {k: v.astype(np.float32) for k, v in p.items()}


def synth_cam(B, seed=0):
    rng = np.random.default_rng(seed + 7919)
    cam = np.stack([rng.uniform(6, 10, B), rng.uniform(-.05, .05, B), rng.uniform(-.05, .05, B)], 1)
    return cam.astype(np.float32)


def synth_images(B, seed=0):
    """Smooth random RGB fields + a little white noise in [0,1] (torch CPU, seeded): unlike pure white noise the
    global-average-pooled CNN features differ visibly from image to image."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(B, 3, 7, 7, generator=g)
    up = F.interpolate(low, size=(224, 224), mode="bilinear", align_corners=False)
    img = 0.5 + 0.22 * up + 0.04 * torch.randn(B, 3, 224, 224, generator=g)
    return img.clamp(0, 1).contiguous()


def synth_generator_input(B, seed=0):
    """[B,6,224,224]: channels 0-2 a grey 'rendered' blob with exact-zero background (3 identical channels, like
    renderer.py:158-166), channels 3-5 an image masked outside a random disc (stand-in for utils/masking.py)."""
    import torch
    g = torch.Generator().manual_seed(seed + 31337)
    img = synth_images(B, seed + 1)
    yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
    c = 112 + 20 * torch.randn(B, 2, generator=g)
    r = 60 + 25 * torch.rand(B, generator=g)
    disc = (((yy[None] - c[:, 0, None, None]) ** 2 + (xx[None] - c[:, 1, None, None]) ** 2) < (r[:, None, None] ** 2)).float()
    shade = synth_images(B, seed + 2)[:, :1] * 0.794
    rendered = (shade * disc[:, None]).repeat(1, 3, 1, 1)
    masked = img * (1 - disc[:, None])
    return torch.cat([rendered, masked], 1).contiguous()


def he_init_(module, seed=0):
    """Seeded He-normal initialisation of every conv / transposed-conv weight of a parameter-holder module (std = sqrt(2 / fan_in) in front
    of a ReLU, sqrt(1 / fan_in) for the linear projections: MobileNetV3 `conv_pwl`, the DepthwiseSeparable block's `conv_pw`, the second
    conv of a ResnetBlock, ConvTranspose2d), BatchNorm left at identity: random-init weights of the reference architecture whose activations
    stay O(1) through all layers (benchmarks; nn.Conv2d's default init shrinks activations ~2.4x per layer, which after 30 layers is
    numerically meaningless)."""
    import torch
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    mods = dict(module.named_modules())
    with torch.no_grad():
        for name, m in mods.items():
            if isinstance(m, nn.ConvTranspose2d):          # weight [Cin, Cout, kh, kw]; stride == kernel => each output sees Cin inputs
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.0 / m.weight.shape[0]) ** 0.5)
                if m.bias is not None:
                    m.bias.zero_()
            elif isinstance(m, nn.Conv2d):
                parent = mods.get(name.rsplit(".", 1)[0]) if "." in name else None
                linear = name.endswith("conv_pwl") or name.endswith("conv_block.5") or \
                    (name.endswith("conv_pw") and getattr(parent, "kind", None) == "ds")
                fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * ((1.0 if linear else 2.0) / fan_in) ** 0.5)
                if m.bias is not None:
                    m.bias.zero_()
    return module


def calibrate_encoder_heads_(enc, device, seed=4242, n_exp=50):
    """Rescale the three linear heads of a random-init SmirkEncoder (already on `device`) so that the regressed FLAME / camera parameters fall
    in the ranges the trained network produces (pose +-0.4, cam scale ~8, |t| < 0.1, shape ~N(0, 0.5), exp ~N(0, 1), jaw[0] in [0, .5], eyelids in
    [0, 1]) — otherwise a random backbone's O(100) features drive FLAME into pathological meshes that no real frame produces (and that cost the
    rasteriser 50x its normal time).  Uses the product's own backbone forward on 16 synthetic frames; benchmark set-up only."""
    import torch
    from smirk_amd.smirk_encoder import features_f32
    g = torch.Generator().manual_seed(seed)
    x = synth_images(16, seed=seed).to(device)
    plan = ((enc.pose_encoder.encoder, enc.pose_encoder.pose_cam_layers[0], [0, 0, 0, 8, 0, 0], [.15, .15, .15, .7, .03, .03]),
            (enc.shape_encoder.encoder, enc.shape_encoder.shape_layers[0], [0.0], [0.5]),
            (enc.expression_encoder.encoder, enc.expression_encoder.expression_layers[0], [0.0] * n_exp + [.5, .5, .2, 0, 0],
             [1.0] * n_exp + [.3, .3, .15, .1, .1]))
    with torch.no_grad():
        for bb, lin, mean, std in plan:
            f = features_f32(bb, bb(x)).float().mean((1, 2)).cpu()                # [16, C] pooled features
            mu, sg = f.mean(0), f.std(0).clamp_min(1e-3 * f.abs().mean().clamp_min(1e-12))
            C, n = f.shape[1], lin.out_features
            mean = torch.tensor(mean, dtype=torch.float32).expand(n) if len(mean) == 1 else torch.tensor(mean, dtype=torch.float32)
            std = torch.tensor(std, dtype=torch.float32).expand(n) if len(std) == 1 else torch.tensor(std, dtype=torch.float32)
            W = torch.randn(n, C, generator=g) / C ** 0.5 / sg[None] * std[:, None]
            lin.weight.copy_(W.to(lin.weight.device))
            lin.bias.copy_((mean - W @ mu).to(lin.bias.device))
    return enc
