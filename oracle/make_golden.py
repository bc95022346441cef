"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Regenerates tests/golden/*.npz by running the REAL reference classes (imported from /root/reference through
oracle/sandbox.py) on seeded synthetic inputs.  Runs only in the build container; the fixtures travel to the GPU box.

    python -m oracle.make_golden            # writes tests/golden/{assets_bundle,flame,flame_grad,render,render_grad,generator,encoder,masking}_golden.npz

What each fixture pins
  flame_golden      reference FLAME.forward (FLAME.py:232-315) outputs, B=4                      -> fully pinned
  flame_grad_golden autograd through the reference FLAME.forward of a fixed random linear loss, B=3 -> fully pinned
  render_grad_golden autograd through the reference Renderer.forward (normals, shading, projection pinned; barycentric backward =
                    autograd of the pytorch3d formula, PARITY UNPINNED)
  generator_golden  reference SmirkGenerator(6,3,32,5).eval() (smirk_generator.py:51-86), B=1    -> fully pinned
  render_golden     reference Renderer.forward (renderer.py:100-207) on top of oracle/raster_ref.c -> glue pinned,
                    rasteriser itself PARITY UNPINNED (pytorch3d not on disk)
  encoder_golden    reference SmirkEncoder.forward heads/clamps (smirk_encoder.py:34-133) on top of
                    oracle/mobilenet_ref.py backbones                                            -> heads pinned,
                    backbone PARITY UNPINNED (timm not on disk)
"""
import os
import tempfile

import numpy as np
import torch

from . import assets as A
from . import generator_ref as G
from . import mobilenet_ref as M
from . import sandbox as S

GOLD = os.path.join(A.REPO, "tests", "golden")


def main():
    assert S.available(), "needs /root/reference"
    A.build_bundle_from_reference(S.REF_ROOT)
    bundle = A.load_bundle()
    d = tempfile.mkdtemp(prefix="smirk_sandbox_")
    A.write_sandbox(d, bundle)
    with S.reference(d) as ref, torch.no_grad():
        # ---- FLAME ------------------------------------------------------------------------------
        p = A.synth_flame_params(4, seed=11)
        fl = ref.FLAME()
        out = fl.forward({k: torch.from_numpy(v) for k, v in p.items()})
        np.savez_compressed(os.path.join(GOLD, "flame_golden.npz"), seed=11,
                            **{"in_" + k: v for k, v in p.items()},
                            **{k: v.numpy() for k, v in out.items()})
        # ---- Renderer ---------------------------------------------------------------------------
        cam = A.synth_cam(4, seed=11)
        rn = ref.Renderer()
        ro = rn.forward(out["vertices"][:2], torch.from_numpy(cam[:2]), landmarks_fan=out["landmarks_fan"][:2],
                        landmarks_mp=out["landmarks_mp"][:2])
        img = ro["rendered_img"].numpy()
        assert np.array_equal(img[:, 0], img[:, 1]) and np.array_equal(img[:, 0], img[:, 2])
        np.savez_compressed(os.path.join(GOLD, "render_golden.npz"), cam=cam[:2],
                            rendered_ch0=img[:, 0], transformed_vertices=ro["transformed_vertices"].numpy(),
                            landmarks_fan=ro["landmarks_fan"].numpy(), landmarks_mp=ro["landmarks_mp"].numpy())
        # ---- Generator --------------------------------------------------------------------------
        sd = G.synth_state_dict()
        g = ref.SmirkGenerator(in_channels=6, out_channels=3, init_features=32, res_blocks=5)
        g.load_state_dict(sd)
        g.eval()
        x = A.synth_generator_input(1, seed=21)
        y = g(x).numpy()
        np.savez_compressed(os.path.join(GOLD, "generator_golden.npz"), seed=21, y_sub4=y[:, :, ::4, ::4],
                            y_sum=np.float64(y.astype(np.float64).sum()), y_sumsq=np.float64((y.astype(np.float64) ** 2).sum()),
                            w_checksum=np.float64(sum(float(v.double().abs().sum()) for v in sd.values())))
        # ---- Encoder ----------------------------------------------------------------------------
        esd = M.synth_encoder_state_dict()
        e = ref.SmirkEncoder()
        e.load_state_dict(esd)
        e.eval()
        img_in = A.synth_images(2, seed=31)
        eo = e(img_in)
        np.savez_compressed(os.path.join(GOLD, "encoder_golden.npz"), seed=31,
                            w_checksum=np.float64(sum(float(v.double().abs().sum()) for v in esd.values())),
                            **{k: v.numpy() for k, v in eo.items()})
        # ---- masking utilities (deterministic parts) -------------------------------
        Mk = ref.masking
        tv = ro["transformed_vertices"]
        faces_t = fl.faces_tensor
        prob = Mk.load_probabilities_per_FLAME_triangle()
        B2 = tv.shape[0]
        fexp = faces_t.expand(B2, -1, -1)
        nrm = ref.render_util.vertex_normals(tv, fexp)
        fnz = ref.render_util.face_vertices(nrm, fexp)[:, :, :, 2].mean(dim=-1)
        w = torch.where(fnz < 0.05, prob.repeat(B2, 1), torch.zeros_like(fnz)) * Mk.triangle_area(ref.render_util.face_vertices(tv, fexp))
        g = torch.Generator().manual_seed(5)
        n_pts = int(0.01 * 224 * 224)
        idx = torch.multinomial(w, n_pts, replacement=True, generator=g)
        u, v = torch.rand(B2 * n_pts, generator=g), torch.rand(B2 * n_pts, generator=g)
        o = (u + v) > 1
        u[o], v[o] = 1 - u[o], 1 - v[o]
        bary = torch.stack((1 - (u + v), u, v), 1).view(B2, n_pts, 3)
        npts, _ = Mk.mesh_based_mask_uniform_faces(tv, faces_t, prob, mask_ratio=0.01,
                                                   coords={"sampled_faces_indices": idx, "barycentric_coords": bary})
        img_m = A.synth_images(B2, seed=41)
        hull = (A.synth_generator_input(B2, seed=41)[:, 3:4] == 0).float()          # a disc, stand-in for the landmark hull mask
        rmask = 1 - (ro["rendered_img"] == 0).all(dim=1, keepdim=True).float()
        pmask = torch.zeros_like(rmask)
        for bi in range(B2):
            pmask[bi, :, npts[bi, :, 1], npts[bi, :, 0]] = 1
        extra = img_m * pmask
        masked = Mk.masking(img_m, hull, extra, 10, rendered_mask=rmask, extra_noise=False, random_mask=0)
        npts2 = torch.flip(npts, [1])
        tp = Mk.transfer_pixels(img_m, npts, npts2)
        np.savez_compressed(os.path.join(GOLD, "masking_golden.npz"), weights=w.numpy(), idx=idx.numpy().astype(np.int32),
                            bary=bary.numpy(), npoints=npts.numpy(), masked_sub2=masked.numpy()[:, :, ::2, ::2],
                            masked_sum=np.float64(masked.double().sum()), transfer_nonzero=np.int64((tp != 0).sum()),
                            transfer_sum=np.float64(tp.double().sum()), img_seed=41)
    # ---- FLAME gradients: autograd through the REAL reference FLAME, B=3 --------------------------
    from .flame_torch_ref import scalar_loss
    with S.reference(d) as ref:
        fl = ref.FLAME()
        p = A.synth_flame_params(3, seed=17)
        tp = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in p.items()}
        loss, _ = scalar_loss(fl.forward(tp), seed=3)
        loss.backward()
        np.savez_compressed(os.path.join(GOLD, "flame_grad_golden.npz"), seed=17, loss_seed=3, loss=np.float64(loss.item()),
                            **{"in_" + k: v for k, v in p.items()}, **{"d_" + k: v.grad.numpy() for k, v in tp.items()})
    # ---- Renderer gradients (§8 f-2): autograd through the REAL reference Renderer.forward; the pytorch3d shim supplies visibility from
    #      oracle/raster_ref.c and differentiable barycentrics from oracle/render_torch_ref.py (third-party part PARITY UNPINNED) ----
    from . import render_torch_ref as RT
    from .flame_ref import FlameRef
    with S.reference(d) as ref:
        rn = ref.Renderer()
        fo = FlameRef(d).forward(A.synth_flame_params(2, seed=23))
        cam = A.synth_cam(2, seed=23)
        leaf = lambda a: torch.from_numpy(np.ascontiguousarray(a)).clone().requires_grad_(True)
        v, c, lf, lm = leaf(fo["vertices"]), leaf(cam), leaf(fo["landmarks_fan"]), leaf(fo["landmarks_mp"])
        o = rn.forward(v, c, landmarks_fan=lf, landmarks_mp=lm)
        loss, _ = RT.scalar_loss(o, seed=29)
        loss.backward()
        np.savez_compressed(os.path.join(GOLD, "render_grad_golden.npz"), loss_seed=29, loss=np.float64(loss.item()),
                            in_vertices=v.detach().numpy(), in_cam=c.detach().numpy(), in_landmarks_fan=lf.detach().numpy(),
                            in_landmarks_mp=lm.detach().numpy(), d_vertices=v.grad.numpy(), d_cam=c.grad.numpy(),
                            d_landmarks_fan=lf.grad.numpy(), d_landmarks_mp=lm.grad.numpy())
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
