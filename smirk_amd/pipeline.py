"""End-to-end per-frame path and its data-parallel sharding.

    encode (SmirkEncoder) -> FLAME -> Renderer -> SmirkGenerator           (demo.py:107-112,167-169; smirk_trainer.py:37-48,94)

Frames are independent, every constant is replicated (~163 MB) and there is NO exchange inside the path, so the batch is
sharded in contiguous slices, one process per GPU; the only collective is an all-gather of the outputs (vertices, rendered and
re-synthesised images) over RCCL/xGMI, issued asynchronously so it overlaps the next batch's compute.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of `total` frames owned by `rank` (remainder spread over the first ranks)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class SmirkPipeline:
    """Holds the four drop-in modules; __call__ runs one batch of frames that are already resident on this GPU."""

    def __init__(self, encoder, flame, renderer, generator=None, face_probabilities=None):
        self.encoder, self.flame, self.renderer, self.generator = encoder, flame, renderer, generator
        self.face_probabilities = face_probabilities      # enables hull_mask= (masking utilities, demo.py:138-165)

    def masked_from_hull(self, img, hull_mask, rn):
        from . import masking as M
        if self.face_probabilities is None:
            raise ValueError("SmirkPipeline(face_probabilities=masking.load_probabilities_per_FLAME_triangle()) is required for hull_mask=")
        return M.demo_masked_image(img, hull_mask, rn['rendered_img'], rn['transformed_vertices'], self.flame.faces_tensor,
                                   self.face_probabilities)

    @torch.no_grad()
    def __call__(self, img, masked_img=None, with_landmarks=True, hull_mask=None):
        enc = self.encoder(img)
        fl = self.flame.forward(enc)
        lm = dict(landmarks_fan=fl['landmarks_fan'], landmarks_mp=fl['landmarks_mp']) if with_landmarks else {}
        rn = self.renderer.forward(fl['vertices'], enc['cam'], **lm)
        out = dict(enc)
        out.update(vertices=fl['vertices'], landmarks_fan_3d=fl['landmarks_fan_3d'], **rn)
        if self.generator is not None:
            if masked_img is None and hull_mask is not None:
                masked_img = self.masked_from_hull(img, hull_mask, rn)
                out['masked_img'] = masked_img
            if masked_img is None:
                raise ValueError("the generator needs the masked image (or hull_mask=) next to the rendering")
            out['reconstructed_img'] = self.generator.forward_pair(rn['rendered_img'], masked_img)     # torch.cat never materialised
        return out


class OverlappedPipeline:
    """Software-pipelines consecutive, independent frame batches over two HIP streams: the latency/bandwidth-bound front
    (encode -> FLAME -> render) of batch i+1 runs concurrently with the MFMA-bound generator of batch i, so the matrix cores and the
    memory system are both kept busy.  Results are identical to SmirkPipeline.__call__ (same kernels, same inputs); only the
    interleaving on the GPU changes.

        run = OverlappedPipeline(pipe)
        for img, masked in batches:
            done = run.submit(img, masked)      # -> outputs of the PREVIOUS batch (None the first time)
        last = run.flush()
    """

    def __init__(self, pipe):
        self.pipe = pipe
        self.front_stream, self.gen_stream = None, None
        self._pending = None                       # (front outputs, masked, event) of the batch whose generator has not run yet

    def _streams(self, device):
        if self.front_stream is None or self.front_stream.device != device:
            self.front_stream, self.gen_stream = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)

    @torch.no_grad()
    def _generate_pending(self):
        if self._pending is None:
            return None
        out, masked, ev = self._pending
        self._pending = None
        caller = torch.cuda.current_stream()
        with torch.cuda.stream(self.gen_stream):
            self.gen_stream.wait_event(ev)
            g = self.pipe.generator
            if isinstance(masked, tuple):                   # ("hull", img, hull_mask): masking utilities run with the generator stage
                _, im, hull = masked
                masked = self.pipe.masked_from_hull(im, hull, out)
                out['masked_img'] = masked
            out['reconstructed_img'] = g.forward_pair(out['rendered_img'], masked)
            for t in (out['rendered_img'], masked):
                t.record_stream(self.gen_stream)
            done = torch.cuda.Event()
            done.record(self.gen_stream)
        caller.wait_event(done)                    # the caller's stream may consume the results
        for t in out.values():
            if torch.is_tensor(t):
                t.record_stream(caller)
        return out

    @torch.no_grad()
    def submit(self, img, masked_img=None, with_landmarks=True, hull_mask=None):
        self._streams(img.device)
        caller = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(caller)                        # inputs were produced on the caller's stream
        prev = self._generate_pending()             # enqueue generator(i-1) first: it is the long pole
        p = self.pipe
        with torch.cuda.stream(self.front_stream):
            self.front_stream.wait_event(ready)
            enc = p.encoder(img)
            fl = p.flame.forward(enc)
            lm = dict(landmarks_fan=fl['landmarks_fan'], landmarks_mp=fl['landmarks_mp']) if with_landmarks else {}
            rn = p.renderer.forward(fl['vertices'], enc['cam'], **lm)
            out = dict(enc)
            out.update(vertices=fl['vertices'], landmarks_fan_3d=fl['landmarks_fan_3d'], **rn)
            ev = torch.cuda.Event()
            ev.record(self.front_stream)
            img.record_stream(self.front_stream)
        self._pending = (out, masked_img if hull_mask is None else ("hull", img, hull_mask), ev)
        return prev

    def flush(self):
        return self._generate_pending()


class OutputGatherer:
    """All-gather of per-rank outputs with equal shard sizes (RCCL on GPUs, gloo in the CPU tests).

    `start()` enqueues the collectives asynchronously into preallocated [world*n, ...] buffers and returns; `wait()` blocks
    until the previous start() has landed.  Typical loop:  out = pipe(x); g.wait(); g.start(out)  — the gather of batch i
    overlaps the compute of batch i+1."""

    KEYS = ('vertices', 'rendered_img', 'reconstructed_img')

    def __init__(self, keys=KEYS, group=None):
        self.keys, self.group = tuple(keys), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bufs, self.pending, self._hold = {}, [], None

    def start(self, outputs):
        self._hold = {k: outputs[k].contiguous() for k in self.keys if k in outputs}
        if self.world == 1:
            self.bufs = dict(self._hold)
            return
        for k, t in self._hold.items():
            shape = (self.world * t.shape[0],) + tuple(t.shape[1:])
            if k not in self.bufs or self.bufs[k].shape != shape or self.bufs[k].device != t.device:
                self.bufs[k] = torch.empty(shape, dtype=t.dtype, device=t.device)
            self.pending.append(dist.all_gather_into_tensor(self.bufs[k], t, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        return self.bufs
