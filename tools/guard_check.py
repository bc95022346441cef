"""Out-of-bounds write detector for the whole path: every buffer the modules allocate through torch.empty (outputs, workspaces) is carved out of a
larger block with GUARD bytes of a known pattern on both sides; after a pass of the pipeline all guards must still hold the pattern.
    python tools/guard_check.py [B]            (GPU box, via gpurun)"""
import os
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import assets as A                      # noqa: E402
from oracle import generator_ref as G               # noqa: E402
from oracle import mobilenet_ref as M               # noqa: E402

GUARD = 2 << 20
PATTERN = 0x5A
_real_empty = torch.empty
_guarded = []
_tag = ["setup"]


def guarded_empty(*size, dtype=None, device=None, **kw):
    if device is None or torch.device(device).type != "cuda" or kw:
        return _real_empty(*size, dtype=dtype, device=device, **kw)
    shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
    dt = dtype or torch.get_default_dtype()
    n = 1
    for s in shape:
        n *= int(s)
    nbytes = n * torch.empty((), dtype=dt).element_size()
    pad = (-nbytes) % 256
    raw = _real_empty(GUARD + nbytes + pad + GUARD, dtype=torch.uint8, device=device)
    raw[:GUARD].fill_(PATTERN)
    raw[GUARD + nbytes:].fill_(PATTERN)
    t = raw[GUARD:GUARD + nbytes].view(dt).view(shape)
    _guarded.append((_tag[0], shape, dt, raw, nbytes))
    return t


def check(label):
    torch.cuda.synchronize()
    bad = 0
    for tag, shape, dt, raw, nbytes in _guarded:
        lo, hi = raw[:GUARD], raw[GUARD + nbytes:]
        for side, g in (("before", lo), ("after", hi)):
            m = g != PATTERN
            if bool(m.any()):
                idx = m.nonzero().flatten()
                off = (idx - GUARD) if side == "before" else idx
                print(f"  OOB WRITE {side} buffer allocated in [{tag}] shape={shape} {dt}: {idx.numel()} bytes, offsets {int(off.min())}..{int(off.max())} "
                      f"relative to the buffer {'start' if side == 'before' else 'end'}", flush=True)
                bad += 1
                g.fill_(PATTERN)
    print(f"{label}: {len(_guarded)} guarded buffers, {bad} corrupted", flush=True)
    return bad


def main():
    from smirk_amd import FLAME, Renderer, SmirkEncoder, SmirkGenerator, masking as MK
    import synthdata as synth
    from smirk_amd.pipeline import OverlappedPipeline, SmirkPipeline
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    sb = tempfile.mkdtemp()
    synth.write_sandbox(sb)
    cwd = os.getcwd(); os.chdir(sb)
    try:
        fl, rn = FLAME().cuda(), Renderer().cuda()
        fp = MK.load_probabilities_per_FLAME_triangle().cuda()
    finally:
        os.chdir(cwd)
    enc = SmirkEncoder(); enc.load_state_dict(M.synth_encoder_state_dict()); enc = enc.cuda().eval()
    gen = SmirkGenerator(6, 3, 32, 5); gen.load_state_dict(G.synth_state_dict()); gen = gen.cuda().eval()
    img = A.synth_images(B, seed=7001).cuda()
    gin = A.synth_generator_input(B, seed=7001).cuda()
    masked = gin[:, 3:].contiguous()
    hull = (gin[:, 3:4] != 0).float().contiguous()
    torch.cuda.synchronize()
    torch.empty = guarded_empty
    try:
        with torch.no_grad():
            for name, fn in (("pose backbone", lambda: enc.pose_encoder(img)), ("shape backbone", lambda: enc.shape_encoder(img)),
                             ("expression backbone", lambda: enc.expression_encoder(img))):
                _tag[0] = name
                fn()
                check(name)
            _tag[0] = "encoder"
            e = enc(img)
            check("encoder (3 streams)")
            _tag[0] = "flame"
            f = fl.forward(e)
            check("FLAME")
            _tag[0] = "renderer"
            r = rn.forward(f["vertices"], e["cam"], landmarks_fan=f["landmarks_fan"], landmarks_mp=f["landmarks_mp"])
            check("renderer")
            _tag[0] = "masking"
            m = MK.demo_masked_image(img, hull, r["rendered_img"], r["transformed_vertices"], fl.faces_tensor, fp)
            check("masking utilities")
            _tag[0] = "generator"
            gen.forward_pair(r["rendered_img"], masked)
            check("generator")
            _tag[0] = "pipeline"
            pipe = SmirkPipeline(enc, fl, rn, gen, face_probabilities=fp)
            run = OverlappedPipeline(pipe)
            for _ in range(3):
                run.submit(img, hull_mask=hull)
            run.flush()
            check("overlapped pipeline x3")
            _tag[0] = "flame-bwd"
    finally:
        torch.empty = _real_empty


if __name__ == "__main__":
    main()
