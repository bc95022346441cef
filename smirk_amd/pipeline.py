"""End-to-end per-frame path and its data-parallel sharding.

    encode (SmirkEncoder) -> FLAME -> Renderer -> SmirkGenerator           (demo.py:107-112,167-169; smirk_trainer.py:37-48,94)

Frames are independent, every constant is replicated (~163 MB) and there is NO exchange inside the path, so the batch is
sharded in contiguous slices, one process per GPU; the only collective is an all-gather of the outputs (vertices, rendered and
re-synthesised images) over RCCL/xGMI, issued asynchronously so it overlaps the next batch's compute.
"""
import os
import warnings

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

    def masked_from_hull(self, img, hull_mask, rn, rng=None):
        from . import masking as M
        if self.face_probabilities is None:
            raise ValueError("SmirkPipeline(face_probabilities=masking.load_probabilities_per_FLAME_triangle()) is required for hull_mask=")
        return M.demo_masked_image(img, hull_mask, rn['rendered_img'], rn['transformed_vertices'], self.flame.faces_tensor,
                                   self.face_probabilities, rng=rng)

    @torch.no_grad()
    def __call__(self, img, masked_img=None, with_landmarks=True, hull_mask=None, mask_rng=None):
        """`mask_rng`: optional masking.PhiloxStream pinning the random draws of the hull_mask= path (default: keyed by torch's seed)."""
        enc = self.encoder(img)
        fl = self.flame.forward(enc)
        lm = dict(landmarks_fan=fl['landmarks_fan'], landmarks_mp=fl['landmarks_mp']) if with_landmarks else {}
        rn = self.renderer.forward(fl['vertices'], enc['cam'], **lm)
        out = dict(enc)
        out.update(vertices=fl['vertices'], landmarks_fan_3d=fl['landmarks_fan_3d'], **rn)
        if self.generator is not None:
            if masked_img is None and hull_mask is not None:
                masked_img = self.masked_from_hull(img, hull_mask, rn, mask_rng)
                out['masked_img'] = masked_img
            if masked_img is None:
                raise ValueError("the generator needs the masked image (or hull_mask=) next to the rendering")
            out['reconstructed_img'] = self.generator.forward_pair(rn['rendered_img'], masked_img)     # torch.cat never materialised
        return out


class OverlappedPipeline:
    """Software-pipelines consecutive, independent frame batches over HIP streams: the latency/bandwidth-bound front (encode -> FLAME -> render)
    of batch i+1 runs concurrently with the MFMA-bound generator of batch i, so the matrix cores and the memory system are both kept busy.
    With `generator_streams=2` consecutive batches' generators additionally alternate between two streams, so that the partially filled last
    round of workgroups of one batch's deep layers (784 / 392 tiles on 256 CUs) and its HBM-bound 224x224 layers overlap with the other batch's
    kernels.  Results are identical to SmirkPipeline.__call__ (same kernels, same inputs); only the interleaving on the GPU changes.

        run = OverlappedPipeline(pipe)
        for img, masked in batches:
            done = run.submit(img, masked)      # -> outputs of an EARLIER batch (None while the pipeline fills), in submission order
        while (last := run.flush()) is not None: ...
    """
    _STREAMS = {}                                  # per (device, role): streams are created once per process, not per pipeline object

    def __init__(self, pipe, generator_streams=1):
        self.pipe = pipe
        self.n_gen = int(generator_streams)
        import smirk_amd
        # streams of one step: front, 3 backbones, n_gen generator streams with a chain side stream each, the collective's stream, the caller's
        n_streams = 6 + 2 * self.n_gen
        if smirk_amd.HW_QUEUES_TOO_LATE or int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < n_streams:
            warnings.warn("OverlappedPipeline runs on %d HIP streams but the HIP runtime multiplexes them onto %s hardware queues (packets of a shared queue "
                          "retire in order; results are unaffected, overlap is lost): export GPU_MAX_HW_QUEUES=16 before the process starts (smirk_amd sets it "
                          "when it is imported before the first HIP call)"
                          % (n_streams, "4 (its default, read before smirk_amd was imported)" if smirk_amd.HW_QUEUES_TOO_LATE
                             else os.environ.get("GPU_MAX_HW_QUEUES")), stacklevel=2)
        self.front_stream, self.gen_streams = None, None
        self._waiting = []                         # front done, generator not enqueued yet: (outputs, masked, front-done event)
        self._running = []                         # generator enqueued: (outputs, generator-done event), oldest first
        self._count = 0

    def _streams(self, device):
        key = (device, self.n_gen)
        if key not in self._STREAMS:
            self._STREAMS[key] = (torch.cuda.Stream(device=device), [torch.cuda.Stream(device=device) for _ in range(self.n_gen)])
        self.front_stream, self.gen_streams = self._STREAMS[key]

    @property
    def gen_stream(self):
        return self.gen_streams[0]

    @torch.no_grad()
    def _start_generators(self):
        """enqueue the generator stage of every batch whose front stages have been enqueued (each on the next generator stream)"""
        while self._waiting:
            out, masked, ev = self._waiting.pop(0)
            gs = self.gen_streams[self._count % self.n_gen]
            self._count += 1
            with torch.cuda.stream(gs):
                gs.wait_event(ev)
                g = self.pipe.generator
                if isinstance(masked, tuple):                   # ("hull", img, hull_mask): masking utilities run with the generator stage
                    _, im, hull, rng = masked
                    masked = self.pipe.masked_from_hull(im, hull, out, rng)
                    out['masked_img'] = masked
                    for t in (out['transformed_vertices'], im, hull):
                        t.record_stream(gs)
                out['reconstructed_img'] = g.forward_pair(out['rendered_img'], masked)
                for t in (out['rendered_img'], masked):
                    t.record_stream(gs)
                done = torch.cuda.Event()
                done.record(gs)
            self._running.append((out, done))

    def _pop_finished(self, keep):
        """hand the oldest batch back once more than `keep` generators are in flight"""
        if len(self._running) <= keep:
            return None
        out, done = self._running.pop(0)
        caller = torch.cuda.current_stream()
        caller.wait_event(done)                    # the caller's stream may consume the results
        for t in out.values():
            if torch.is_tensor(t):
                t.record_stream(caller)
        return out

    @torch.no_grad()
    def submit(self, img, masked_img=None, with_landmarks=True, hull_mask=None, mask_rng=None):
        self._streams(img.device)
        caller = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(caller)                        # inputs were produced on the caller's stream
        self._start_generators()                    # enqueue generator(i-1) first: it is the long pole
        prev = self._pop_finished(self.n_gen - 1)
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
        self._waiting.append((out, masked_img if hull_mask is None else ("hull", img, hull_mask, mask_rng), ev))
        return prev

    def flush(self):
        """next finished batch in submission order, None when the pipeline is empty"""
        self._start_generators()
        return self._pop_finished(0)


class RotatingPipeline:
    """Consecutive, independent frame batches of a generator-less SmirkPipeline (encode -> FLAME -> render: demo.py:107-112 in a loop) on `lanes` HIP streams in
    rotation, so that the thinly occupied tail of batch i (raster_tile: 1.5 workgroups per CU at 256 frames) runs while the backbones of batch i+1 have already
    started.  Same kernels, same inputs, results identical to SmirkPipeline.__call__; only the interleaving on the GPU changes.

        run = RotatingPipeline(pipe, lanes=2)
        for img in batches:
            done = run.submit(img)              # -> outputs of the batch submitted `lanes - 1` calls earlier (None while the pipeline fills)
        while (last := run.flush()) is not None: ...
    """
    _STREAMS = {}

    def __init__(self, pipe, lanes=2):
        if pipe.generator is not None:
            raise ValueError("RotatingPipeline is the generator-less schedule; use OverlappedPipeline for encode -> ... -> generate")
        self.pipe, self.lanes = pipe, max(1, int(lanes))
        self._running, self._count = [], 0

    def _pop(self, keep):
        if len(self._running) <= keep:
            return None
        out, done = self._running.pop(0)
        caller = torch.cuda.current_stream()
        caller.wait_event(done)
        for t in out.values():
            if torch.is_tensor(t):
                t.record_stream(caller)
        return out

    @torch.no_grad()
    def submit(self, img, with_landmarks=True):
        key = (img.device, self.lanes)
        if key not in self._STREAMS:
            self._STREAMS[key] = [torch.cuda.Stream(device=img.device) for _ in range(self.lanes)]
        lane = self._STREAMS[key][self._count % self.lanes]
        self._count += 1
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())               # the input was produced on the caller's stream
        with torch.cuda.stream(lane):
            lane.wait_event(ready)
            out = self.pipe(img, with_landmarks=with_landmarks)
            img.record_stream(lane)
            done = torch.cuda.Event()
            done.record(lane)
        self._running.append((out, done))
        return self._pop(self.lanes - 1)

    def flush(self):
        """next finished batch in submission order, None when the pipeline is empty"""
        return self._pop(0)


class OutputGatherer:
    """All-gather of per-rank outputs with equal shard sizes (RCCL on GPUs, gloo in the CPU tests).

    `start()` enqueues the collectives asynchronously into preallocated [world*n, ...] buffers and returns; `wait()` blocks
    until the previous start() has landed.  Typical loop:  out = pipe(x); g.wait(); g.start(out)  — the gather of batch i
    overlaps the compute of batch i+1.

    Stream contract: the collective runs on the process group's own stream.  c10d makes that stream wait for the CURRENT stream at the
    time of start() (so outputs produced on the current stream are complete before they are sent) and `wait()` makes the current stream
    wait for the collective; inputs and outputs are `record_stream`-ed on the communication stream by c10d, and `_hold` keeps the inputs
    referenced until wait().  With `force_collective=True` a world of ONE still issues the real collective (a single MI355X then exercises
    RCCL's all_gather_into_tensor end to end: tests/test_rccl_gpu.py); the default short-circuits it."""

    KEYS = ('vertices', 'rendered_img', 'reconstructed_img')

    def __init__(self, keys=KEYS, group=None, force_collective=False):
        self.keys, self.group = tuple(keys), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(force_collective) and dist.is_initialized()
        self.bufs, self.pending, self._hold, self._pool = {}, [], None, {}

    def start(self, outputs):
        self._hold = {k: outputs[k].contiguous() for k in self.keys if k in outputs}
        if self.world == 1 and not self.force:
            self.bufs = dict(self._hold)
            return
        for k, t in self._hold.items():
            shape = (self.world * t.shape[0],) + tuple(t.shape[1:])
            pool = self._pool.setdefault((k, shape, t.device), None)                # one buffer per distinct micro-batch shape: no allocator churn
            if pool is None:
                pool = self._pool[(k, shape, t.device)] = torch.empty(shape, dtype=t.dtype, device=t.device)
            self.bufs[k] = pool
            self.pending.append(dist.all_gather_into_tensor(pool, t, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        return self.bufs
