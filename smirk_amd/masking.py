"""MI355X drop-in for the reference masking utilities (src/utils/masking.py) — the step between Renderer and SmirkGenerator
(demo.py:146-167, smirk_trainer.py:76-93,268-293).  Same function names, arguments and return values:

    load_probabilities_per_FLAME_triangle, triangle_area, random_barycentric, masking, point2ind, transfer_pixels,
    mesh_based_mask_uniform_faces

The per-frame work runs in libsmirk_hip.so (smirk_amd/csrc/masking.hip): full-mesh vertex normals by CSR gather, per-triangle sampling
weights, multinomial-with-replacement sampling (LDS prefix-sum CDF + Philox + binary search) fused with the barycentric draw, the
landmark gather, pixel quantisation, separable max-pool dilations and the final composition.  Random numbers come from a counter-based
Philox stream keyed by torch's global seed (`torch.manual_seed`) and a call counter — reproducible, but (like the reference's own CPU
vs CUDA runs) not the same numbers as torch's generators; pass `coords=` for deterministic geometry, exactly as the reference allows.
"""
import numpy as np
import torch

from . import _lib as L
from .renderer import normal_csr

class PhiloxStream:
    """Explicit state of this module's random draws: key `seed`, running counter `offset`.  Every kernel that draws takes
    (seed, offset) and consumes `n_draws` counters; passing the same PhiloxStream(seed, offset) to two runs makes them draw the same
    numbers whatever else touched torch's generators in between (tests pin the bench schedule this way)."""

    def __init__(self, seed, offset=0):
        self.seed, self.offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset)

    def __call__(self, n_draws):
        off = self.offset
        self.offset += int(n_draws)
        return self.seed, off


def _rng(n_draws, stream=None):
    """(seed, offset) of the Philox counters for one call.  Default: the key is torch's global seed and the counter base is drawn from
    torch's default (CPU) generator, so `torch.manual_seed(s)` makes a whole sequence of calls reproducible while successive calls differ;
    with an explicit PhiloxStream the draw does not touch torch's generators at all."""
    if stream is not None:
        return stream(n_draws)
    seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    off = int(torch.randint(0, 2 ** 62, (1,)).item())
    return seed, off


def load_probabilities_per_FLAME_triangle():
    """masking.py:10-36: per-triangle sampling prior from the FLAME region lists (host-side, cwd-relative asset like the reference)."""
    tri = np.load('assets/FLAME_masks/FLAME_masks_triangles.npy', allow_pickle=True).item()
    area_weights = {'neck': 0.0, 'right_eyeball': 0.0, 'right_ear': 0.0, 'lips': 0.5, 'nose': 0.5, 'left_ear': 0.0, 'eye_region': 1.0,
                    'forehead': 1.0, 'left_eye_region': 1.0, 'right_eye_region': 1.0, 'face_clean': 1.0, 'cleaner_lips': 1.0}
    p = torch.zeros(9976)
    for area, wgt in area_weights.items():
        p[tri[area]] = wgt
    return p


def triangle_area(vertices):
    """masking.py:39-47 (shoelace area in the xy plane; small host/device utility kept for API parity)."""
    x1, y1 = vertices[..., 0, 0], vertices[..., 0, 1]
    x2, y2 = vertices[..., 1, 0], vertices[..., 1, 1]
    x3, y3 = vertices[..., 2, 0], vertices[..., 2, 1]
    return 0.5 * torch.abs(x1 * y2 + x2 * y3 + x3 * y1 - x2 * y1 - x3 * y2 - x1 * y3)


def random_barycentric(num=1):
    """masking.py:51-68 (API parity; the hot path draws barycentrics inside smirk_sample_faces)."""
    u, v = torch.rand(num), torch.rand(num)
    out = (u + v) > 1
    u[out], v[out] = 1 - u[out], 1 - v[out]
    return torch.stack((1 - (u + v), u, v), dim=1)


def _maxpool_sq(x, radius, complement):
    B, _, H, W = x.shape
    tmp, out = torch.empty_like(x), torch.empty_like(x)
    L.check(L.lib().smirk_maxpool_sq(L.ptr(x), L.ptr(tmp), L.ptr(out), B, H, W, radius, int(complement), L.stream_ptr()))
    return out


def masking(img, mask, extra_points, wr=15, rendered_mask=None, extra_noise=True, random_mask=0.01, _noise_mult=None, _random_field=None,
            _pmask=None, _rng_stream=None):
    """masking.py:71-102.  img [B,C,H,W], mask / rendered_mask [B,1,H,W], extra_points [B,C,H,W] -> masked_img [B,C,H,W].
    `_noise_mult` ([B,C,H,W]) and `_random_field` ([B,1,H,W] of 0/1 patch centres) override the random draws (tests);
    `_pmask` ([B,1,H,W]) with extra_points=None fuses the caller's `extra_points = img * pmask` (demo.py:163)."""
    img, mask = L.as_f32c(img), L.as_f32c(mask)
    extra_points = None if extra_points is None else L.as_f32c(extra_points)
    B, C, H, W = img.shape
    # the reference's own call site passes a 3-D hull mask [1,H,W] (demo.py:161, demo_video.py): F.max_pool2d's unbatched mode plus
    # broadcasting make that work there; normalise to [B or 1, 1, H, W] here
    if mask.dim() == 3:
        mask = mask.unsqueeze(0) if mask.shape[0] == 1 else mask.unsqueeze(1)          # [1,H,W] -> [1,1,H,W] (broadcast over the batch); [B,H,W] -> [B,1,H,W]
    if rendered_mask is not None and rendered_mask.dim() == 3:
        rendered_mask = rendered_mask.unsqueeze(0) if rendered_mask.shape[0] == 1 else rendered_mask.unsqueeze(1)
    lib, st = L.lib(), L.stream_ptr()
    if mask.shape[0] != B:
        mask = mask.expand(B, -1, -1, -1).contiguous()
    mask_d = _maxpool_sq(mask, wr, complement=3)                             # 1 - maxpool(1 - mask, 2wr+1)
    rm = None if rendered_mask is None else L.as_f32c(rendered_mask)
    if rm is not None and rm.shape[0] != B:
        rm = rm.expand(B, -1, -1, -1).contiguous()
    keep = None
    if random_mask > 0 or _random_field is not None:
        if _random_field is None:
            field = torch.empty(B, 1, H, W, device=img.device)
            seed, off = _rng((B * H * W + 3) // 4, _rng_stream)
            L.check(lib.smirk_bernoulli_field(L.ptr(field), field.numel(), float(random_mask), seed, off, st))
        else:
            field = L.as_f32c(_random_field)
        keep = _maxpool_sq(field, 5, complement=2)                           # 1 - maxpool(centres, 11): zero inside the 11x11 patches
    out = torch.empty_like(img)
    noise = None if _noise_mult is None else L.as_f32c(_noise_mult)
    seed, off = _rng(img.numel(), _rng_stream) if (extra_noise and noise is None) else (0, 0)
    pm = None if _pmask is None else L.as_f32c(_pmask)
    L.check(lib.smirk_masking_compose(L.ptr(img), L.ptr(mask_d), L.ptr(rm, allow_none=True), L.ptr(extra_points, allow_none=True),
                                      L.ptr(pm, allow_none=True), L.ptr(keep, allow_none=True), L.ptr(noise, allow_none=True), B, C, H, W,
                                      int(bool(extra_noise) and noise is None), seed, off, L.ptr(out), st))
    return out


def point2ind(npoints, H):
    """masking.py:105-113 (index arithmetic on a small tensor; device-side plumbing)."""
    npoints = (npoints * (H // 2) + H // 2).long()
    npoints[..., 1] = torch.clamp(npoints[..., 1], 0, H - 1)
    npoints[..., 0] = torch.clamp(npoints[..., 0], 0, H - 1)
    return npoints


def transfer_pixels(img, points1, points2, rbound=None):
    """masking.py:116-129: retained[b,:,p2y,p2x] = img[b,:,p1y,p1x]; duplicate targets resolve to the highest point index."""
    img = L.as_f32c(img)
    B, C, H, W = img.shape
    p1, p2 = points1.long().contiguous(), points2.long().contiguous()
    Lp = p1.shape[1]
    if p1.shape[-1] == 2:                                                     # accept [B,L,2] like the reference's indexing does
        z = torch.zeros(B, Lp, 1, dtype=torch.long, device=p1.device)
        p1, p2 = torch.cat([p1, z], -1).contiguous(), torch.cat([p2, z], -1).contiguous()
    rb = None if rbound is None else rbound.long().contiguous()
    ws = torch.empty(B, H, W, dtype=torch.int32, device=img.device)
    out = torch.empty_like(img)
    L.check(L.lib().smirk_transfer_pixels(L.ptr(img), L.ptr(p1, torch.int64), L.ptr(p2, torch.int64), L.ptr(rb, torch.int64, True),
                                          B, C, Lp, H, W, L.ptr(ws, torch.int32), L.ptr(out), L.stream_ptr()))
    return out


_mesh_cache = {}


def _full_mesh(flame_faces):
    key = (flame_faces.data_ptr(), flame_faces.device, tuple(flame_faces.shape))
    if key not in _mesh_cache:
        f = flame_faces.detach().cpu().numpy().astype(np.int64)
        nv = int(f.max()) + 1
        ptr, nf, nc = normal_csr(f, nv)
        dev = flame_faces.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        bufs = dict(faces=t(f.astype(np.int32)), ptr=t(ptr), nf=t(nf), nc=t(nc))
        m = L.SmirkRenderMesh()
        m.V, m.Vf, m.Ff, m.nnz = nv, nv, f.shape[0], int(nf.size)
        I = torch.int32
        m.keep = None
        m.faces, m.nrm_ptr, m.nrm_face, m.nrm_corner = L.ptr(bufs['faces'], I), L.ptr(bufs['ptr'], I), L.ptr(bufs['nf'], I), L.ptr(bufs['nc'], I)
        _mesh_cache[key] = (m, bufs, nv, f.shape[0])
    return _mesh_cache[key]


def mesh_based_mask_uniform_faces(flame_trans_verts, flame_faces, face_probabilities, mask_ratio=0.1, coords=None, IMAGE_SIZE=224,
                                  _rng_stream=None):
    """masking.py:132-181: sample `int(mask_ratio * IMAGE_SIZE**2)` points on the visible, region-weighted FLAME surface and return
    their integer pixel coordinates [B,N,3] (x, y, z) plus {'sampled_faces_indices' [B,N] int64, 'barycentric_coords' [B,N,3]}."""
    tv = L.as_f32c(flame_trans_verts)
    B, dev = tv.shape[0], tv.device
    n = int(mask_ratio * IMAGE_SIZE * IMAGE_SIZE)
    lib, st = L.lib(), L.stream_ptr()
    mesh, bufs, V, F = _full_mesh(flame_faces)
    if tv.shape[1] != V:
        raise L.SmirkHipError("vertex count does not match flame_faces")
    if coords is None:
        normals = torch.empty(B, V, 3, device=dev)
        L.check(lib.smirk_vertex_normals(mesh, B, L.ptr(tv), L.ptr(normals), st))
        prob = L.as_f32c(face_probabilities.to(dev))
        wts = torch.empty(B, F, device=dev)
        L.check(lib.smirk_mask_face_weights(L.ptr(tv), L.ptr(normals), L.ptr(bufs['faces'], torch.int32), L.ptr(prob), B, V, F, L.ptr(wts), st))
        idx = torch.empty(B, n, dtype=torch.int32, device=dev)
        bary = torch.empty(B, n, 3, device=dev)
        seed, off = _rng(B * n, _rng_stream)
        L.check(lib.smirk_sample_faces(L.ptr(wts), B, F, n, seed, off, L.ptr(idx, torch.int32), L.ptr(bary), st))
    else:
        idx = coords['sampled_faces_indices'].to(torch.int32).contiguous()
        bary = L.as_f32c(coords['barycentric_coords'])
        n = idx.shape[1]
    pts = torch.empty(B, n, 3, device=dev)
    L.check(lib.smirk_vertices2landmarks(L.ptr(tv), B, V, L.ptr(bufs['faces'], torch.int32), L.ptr(idx, torch.int32), L.ptr(bary), n, L.ptr(pts), st))
    npoints = torch.empty(B, n, 3, dtype=torch.int64, device=dev)
    L.check(lib.smirk_points_to_pixels(L.ptr(pts), B, n, IMAGE_SIZE, L.ptr(npoints, torch.int64), st))
    return npoints, {'sampled_faces_indices': idx.long(), 'barycentric_coords': bary}


def rendered_mask_of(rendered_img):
    """rendered_mask = 1 - (rendered_img == 0).all(dim=1, keepdim=True).float()   (demo.py:146, smirk_trainer.py:79)."""
    x = L.as_f32c(rendered_img)
    B, C, H, W = x.shape
    out = torch.empty(B, 1, H, W, device=x.device)
    L.check(L.lib().smirk_rendered_mask(L.ptr(x), B, C, H, W, L.ptr(out), L.stream_ptr()))
    return out


def points_mask(npoints, H, W, rbound=None):
    """pmask[b, :, y, x] = 1 at the first rbound[b] sampled points (demo.py:153-159) -> [B,1,H,W]."""
    p = npoints.long().contiguous()
    B, Lp = p.shape[:2]
    rb = None if rbound is None else rbound.long().contiguous()
    out = torch.empty(B, 1, H, W, device=p.device)
    L.check(L.lib().smirk_scatter_points_mask(L.ptr(p, torch.int64), L.ptr(rb, torch.int64, True), B, Lp, H, W, L.ptr(out), L.stream_ptr()))
    return out


def demo_masked_image(img, hull_mask, rendered_img, transformed_vertices, flame_faces, face_probabilities,
                      mask_ratio=0.01, mask_ratio_mul=5, mask_dilation_radius=10, rng=None):
    """The block demo.py:138-165 runs between Renderer and SmirkGenerator, on the GPU: rendered mask, mesh-based point sampling with a
    random per-image budget, point mask, masking().  Returns masked_img [B,3,H,W].  `rng`: an optional PhiloxStream that pins every
    draw of the block (sampled faces + barycentrics, per-image budgets, patch centres, point noise) independently of torch's generators."""
    B, _, H, W = img.shape
    rmask = rendered_mask_of(rendered_img)
    npoints, _ = mesh_based_mask_uniform_faces(transformed_vertices, flame_faces, face_probabilities, mask_ratio=mask_ratio * mask_ratio_mul,
                                               IMAGE_SIZE=H, _rng_stream=rng)
    # demo.py:154-156 draws rsing / rscale per image on the host and copies the budgets over; here they are drawn on the device (Philox keyed
    # like the other draws of this module) so the step has no host round trip and can be captured in a hipGraph
    rbound = torch.empty(B, dtype=torch.int64, device=img.device)
    seed, off = _rng(B, rng)
    L.check(L.lib().smirk_random_point_budget(L.ptr(rbound, torch.int64), B, int(npoints.size(1)), float(mask_ratio_mul), seed, off, L.stream_ptr()))
    pmask = points_mask(npoints, H, W, rbound)
    return masking(img, hull_mask, None, mask_dilation_radius, rendered_mask=rmask, _pmask=pmask, _rng_stream=rng)
