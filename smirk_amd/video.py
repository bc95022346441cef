"""Batched, streamed version of the reference's video loop (demo_video.py:107-214).

The reference handles one frame at a time: cv2 decode -> CPU mediapipe -> skimage crop/warp -> hot path at batch 1 -> numpy grid ->
cv2 encode.  Here N decoded frames are staged in pinned host memory, uploaded once, and everything between "uint8 BGR frame" and
"uint8 BGR output grid" runs on the GPU (csrc/video.hip + the hot path), software-pipelined over three HIP streams:

    copy-in stream   H2D of batch i+1            (pinned -> HBM)
    compute stream   warp/crop -> encoder -> FLAME -> renderer -> [masking -> generator] -> grid composition of batch i
    copy-out stream  D2H of batch i-1            (HBM -> pinned), frames handed to the caller in order

Decoding, the mediapipe landmark detector and encoding stay with the caller (third-party CPU components, out of scope): `run()` takes an
iterable of frames and an iterable of landmark arrays (what utils/mediapipe_utils.run_mediapipe returns) and yields output frames.
"""
import collections

import numpy as np
import torch

from . import _lib as L
from .pipeline import SmirkPipeline

IMAGE_SIZE = 224                      # demo_video.py:57


def crop_transform(landmarks, scale=1.0, image_size=IMAGE_SIZE):
    """demo_video.py:16-36 crop_face: similarity transform (3x3, frame (x,y) -> crop (x,y)) that maps the landmark box, enlarged by
    `scale`, onto the crop.  The source points are three corners of an axis-aligned square, so the least-squares similarity skimage
    estimates is the exact scale + translation written here in closed form."""
    lm = np.asarray(landmarks, np.float64)
    left, right, top, bottom = lm[:, 0].min(), lm[:, 0].max(), lm[:, 1].min(), lm[:, 1].max()
    old_size = (right - left + bottom - top) / 2
    cx, cy = right - (right - left) / 2.0, bottom - (bottom - top) / 2.0
    size = int(old_size * scale)
    if size <= 0:
        raise ValueError("degenerate landmark box")
    s = (image_size - 1) / size
    return np.array([[s, 0.0, -s * (cx - size / 2)], [0.0, s, -s * (cy - size / 2)], [0.0, 0.0, 1.0]])


class _Slot:
    def __init__(self):
        self.key = None
        self.used = False


class VideoPipeline:
    """demo_video.py's frame loop as a batched stream.

        vp = VideoPipeline(encoder, flame, renderer, generator, face_probabilities, batch_size=32, crop=True, use_smirk_generator=True)
        for out_frame in vp.run(frames, landmarks):          # uint8 [H_out, W_out, 3] BGR, same order as the input
            writer.write(out_frame)

    Flags mirror the script's: `crop` (--crop), `use_smirk_generator`, `render_orig`.  Output width is 2 or 3 panels
    (demo_video.py:91-99)."""

    def __init__(self, smirk_encoder, flame, renderer, smirk_generator=None, face_probabilities=None, batch_size=32, crop=False,
                 use_smirk_generator=False, render_orig=False, depth=2, device='cuda'):
        if use_smirk_generator and (smirk_generator is None or face_probabilities is None):
            raise ValueError("use_smirk_generator needs smirk_generator and face_probabilities")
        self.pipe = SmirkPipeline(smirk_encoder, flame, renderer, smirk_generator if use_smirk_generator else None, face_probabilities)
        self.N, self.crop, self.use_gen, self.render_orig = int(batch_size), bool(crop), bool(use_smirk_generator), bool(render_orig)
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self._slots = [_Slot() for _ in range(self.depth)]
        self._in_stream = self._out_stream = None

    # ---- buffers -----------------------------------------------------------------------------------------------------------------
    def _prepare(self, slot, Hv, Wv, L_pts):
        key = (Hv, Wv, L_pts)
        if slot.key == key:
            return
        N, dev = self.N, self.device
        S = IMAGE_SIZE
        panels = 3 if self.use_gen else 2
        Ho, Wo = (Hv, Wv) if self.render_orig else (S, S)
        pin = lambda *s, dtype: torch.empty(*s, dtype=dtype).pin_memory()
        slot.h_frames, slot.d_frames = pin(N, Hv, Wv, 3, dtype=torch.uint8), torch.empty(N, Hv, Wv, 3, dtype=torch.uint8, device=dev)
        slot.h_mats, slot.d_mats = pin(N, 2, 6, dtype=torch.float64), torch.empty(N, 2, 6, dtype=torch.float64, device=dev)
        slot.h_lmk = pin(N, max(L_pts, 1), 2, dtype=torch.float32)
        slot.d_lmk = torch.empty(N, max(L_pts, 1), 2, dtype=torch.float32, device=dev)
        slot.d_grid = torch.empty(N, Ho, Wo * panels, 3, dtype=torch.uint8, device=dev)
        slot.h_grid = pin(N, Ho, Wo * panels, 3, dtype=torch.uint8)
        slot.ev_in, slot.ev_done, slot.ev_out = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        slot.key, slot.out_hw, slot.panels, slot.used = key, (Ho, Wo), panels, False

    # ---- one batch ---------------------------------------------------------------------------------------------------------------
    def _submit(self, slot, frames, lmks):
        n = len(frames)
        Hv, Wv = frames[0].shape[:2]
        need_lmk = self.crop or self.use_gen
        if need_lmk and not all(l is not None for l in lmks):
            # demo_video.py:116-119,177-179: the script exits when mediapipe finds nothing and it has to crop / build the hull mask
            raise ValueError("landmarks are required for every frame when crop / use_smirk_generator is set")
        # without crop / generator the landmarks are never consumed and the reference tolerates frames where mediapipe found nothing
        # (demo_video.py:110-119): nothing is staged then, whatever mix of None / arrays the batch holds
        L_pts = int(np.asarray(lmks[0]).shape[0]) if need_lmk else 0
        self._prepare(slot, Hv, Wv, L_pts)
        hf = slot.h_frames.numpy()
        for i, f in enumerate(frames):
            if f.shape != (Hv, Wv, 3) or f.dtype != np.uint8:
                raise ValueError("all frames of a stream must be uint8 [H, W, 3] of one resolution")
            hf[i] = f
        hm, hl = slot.h_mats.numpy(), slot.h_lmk.numpy()
        for i in range(n):
            if self.crop:
                kpt = np.asarray(lmks[i], np.float64)[..., :2]
                T = crop_transform(kpt, scale=1.4, image_size=IMAGE_SIZE)                          # demo_video.py:122
                hm[i, 0] = np.linalg.inv(T)[:2].reshape(6)      # crop (col,row) -> frame (x,y): warp(image, tform.inverse)
                hm[i, 1] = T[:2].reshape(6)                     # frame (col,row) -> crop (x,y): warp(rendered, tform)
                ck = (T @ np.hstack([kpt, np.ones([kpt.shape[0], 1])]).T).T[:, :2]                 # demo_video.py:126-127
            else:
                ck = np.asarray(lmks[i], np.float64)[..., :2] if need_lmk else None
            if ck is not None:
                hl[i] = ck.astype(np.int32)                     # create_mask's landmarks.astype(np.int32)
        for i in range(n, self.N):                              # ragged last batch: replicate the last frame, dropped on output
            hf[i] = hf[n - 1]; hm[i] = hm[n - 1]; hl[i] = hl[n - 1]
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._in_stream):
            if slot.used:                                       # the slot's device buffers: wait for ITS previous batch only, so this
                self._in_stream.wait_event(slot.ev_done)        # upload overlaps the compute of the batch in between
            slot.d_frames.copy_(slot.h_frames, non_blocking=True)
            slot.d_mats.copy_(slot.h_mats, non_blocking=True)
            slot.d_lmk.copy_(slot.h_lmk, non_blocking=True)
            slot.ev_in.record(self._in_stream)
        cur.wait_event(slot.ev_in)
        self._compute(slot, Hv, Wv, L_pts)
        slot.ev_done.record(cur)
        with torch.cuda.stream(self._out_stream):
            self._out_stream.wait_event(slot.ev_done)
            slot.h_grid.copy_(slot.d_grid, non_blocking=True)
            slot.ev_out.record(self._out_stream)
        slot.n, slot.used = n, True

    @torch.no_grad()
    def _compute(self, slot, Hv, Wv, L_pts):
        lib, P, st = L.lib(), L.ptr, L.stream_ptr()
        N, S, dev = self.N, IMAGE_SIZE, self.device
        U8, F64 = torch.uint8, torch.float64
        img = torch.empty(N, 3, S, S, device=dev)
        inv = slot.d_mats[:, 0].contiguous()
        if self.crop:
            L.check(lib.smirk_warp_affine_u8(P(slot.d_frames, U8), N, Hv, Wv, P(inv, F64), S, S, 1, P(img), None, st))
        else:                                                    # cvtColor + cv2.resize(…, (224, 224)) of the whole frame
            L.check(lib.smirk_resize_linear_u8(P(slot.d_frames, U8), N, Hv, Wv, S, S, 1, P(img), None, st))
        hull = None
        if self.use_gen:
            hull = torch.empty(N, 1, S, S, device=dev)
            L.check(lib.smirk_hull_mask(P(slot.d_lmk), N, L_pts, 2, S, S, P(hull), st))
        out = self.pipe(img, hull_mask=hull)
        panels = [img, out['rendered_img']] + ([out['reconstructed_img']] if self.use_gen else [])
        Ho, Wo = slot.out_hw
        gw = Wo * slot.panels
        if not self.render_orig:
            for j, p in enumerate(panels):
                L.check(lib.smirk_f32_nchw_to_u8_grid(P(L.as_f32c(p)), N, S, S, 1, P(slot.d_grid, U8), gw, j * Wo, st))
            return
        full = torch.empty(N, 3, Hv, Wv, device=dev)             # demo_video.py:169
        L.check(lib.smirk_u8_hwc_to_f32_nchw(P(slot.d_frames, U8), N, Hv, Wv, 1, P(full), st))
        L.check(lib.smirk_f32_nchw_to_u8_grid(P(full), N, Hv, Wv, 1, P(slot.d_grid, U8), gw, 0, st))
        fwd = slot.d_mats[:, 1].contiguous()
        tmp8 = torch.empty(N, S, S, 3, dtype=U8, device=dev)
        big = torch.empty(N, 3, Hv, Wv, device=dev)
        for j, p in enumerate(panels[1:], start=1):
            p = L.as_f32c(p)
            if self.crop:                                         # demo_video.py:162-165,202-205
                L.check(lib.smirk_f32_nchw_to_u8_grid(P(p), N, S, S, 0, P(tmp8, U8), S, 0, st))
                L.check(lib.smirk_warp_affine_u8(P(tmp8, U8), N, S, S, P(fwd, F64), Hv, Wv, 0, P(big), None, st))
            else:                                                 # demo_video.py:167,207
                L.check(lib.smirk_interp_bilinear_f32(P(p), N * 3, S, S, Hv, Wv, P(big), st))
            L.check(lib.smirk_f32_nchw_to_u8_grid(P(big), N, Hv, Wv, 1, P(slot.d_grid, U8), gw, j * Wo, st))

    def _drain(self, slot):
        slot.ev_out.synchronize()
        g = slot.h_grid.numpy()
        for i in range(slot.n):
            yield g[i].copy()

    # ---- the stream ----------------------------------------------------------------------------------------------------------------
    def run(self, frames, landmarks=None):
        """frames: iterable of uint8 [H,W,3] BGR arrays (cv2.VideoCapture.read order); landmarks: parallel iterable of [L,>=2] arrays or
        None entries (run_mediapipe's result per frame).  Yields one uint8 BGR grid per input frame, in order."""
        if self._in_stream is None:
            self._in_stream, self._out_stream = torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)
        lm_iter = iter(landmarks) if landmarks is not None else None
        pending = collections.deque()
        k = 0
        bf, bl = [], []

        def flush_batch():
            nonlocal k, bf, bl
            slot = self._slots[k % self.depth]
            k += 1
            self._submit(slot, bf, bl)
            pending.append(slot)
            bf, bl = [], []

        with torch.cuda.device(self.device):
            for f in frames:
                bf.append(np.asarray(f))
                bl.append(next(lm_iter) if lm_iter is not None else None)
                if len(bf) == self.N:
                    if len(pending) == self.depth:                # the slot about to be reused must be drained first
                        yield from self._drain(pending.popleft())
                    flush_batch()
            if bf:
                if len(pending) == self.depth:
                    yield from self._drain(pending.popleft())
                flush_batch()
            while pending:
                yield from self._drain(pending.popleft())
