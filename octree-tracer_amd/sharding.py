"""Multi-GPU framebuffer sharding (no reference counterpart: the reference is single-device).

Primary rays are independent and the node array is read-only, so the frame is cut into
tile_w x tile_h tiles dealt round-robin to ranks (tile t belongs to rank t % world); every rank
holds a full replica of the node array.  The only data-path exchange is ONE gather of hit records
to rank 0 at frame end (torch.distributed: RCCL over xGMI on the GPU box, gloo in CPU tests),
after which rank 0 un-permutes tiles into the row-major frame.
"""
import torch
import torch.distributed as dist


def local_tile_count(width, height, tile_w, tile_h, rank, world):
    n = (width // tile_w) * (height // tile_h)
    return max(0, (n - rank + world - 1) // world)


def padded_tile_count(width, height, tile_w, tile_h, world):
    """every rank's gather buffer holds this many tiles (rank 0 has the most)"""
    return local_tile_count(width, height, tile_w, tile_h, 0, world)


def gather_frame(local, rank, world, dst=0, group=None):
    """local: [n_pad_tiles, tile_h * tile_w, 4] int32 hit records of this rank.
    Returns [world, n_pad_tiles, tile_h * tile_w, 4] on dst, None elsewhere.  One collective."""
    if world == 1:
        return local.unsqueeze(0)
    if rank == dst:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, list(out.unbind(0)), dst=dst, group=group)
        return out
    dist.gather(local, None, dst=dst, group=group)
    return None


def pack_records(records):
    """[..., 4] hit records -> [..., 3] wire records: word 3 (packed_normal) repeats bits 17..22 of word 2."""
    return records[..., :3].contiguous()


def unpack_records(wire):
    """inverse of pack_records"""
    return torch.cat([wire, ((wire[..., 2:3] >> 17) & 63)], dim=-1)


def assemble_frame(gathered, width, height, tile_w, tile_h):
    """[world, n_pad, tile_h*tile_w, 4] (rank r, slot k holds tile r + k*world) -> [H, W, 4].  Wire records
    (last dimension 3) are expanded on the way."""
    if gathered.shape[-1] == 3:
        gathered = unpack_records(gathered)
    world, n_pad, c = gathered.shape[0], gathered.shape[1], gathered.shape[-1]  # c: 4 (records) or 1 (RGBA8 colour)
    tiles_x, tiles_y = width // tile_w, height // tile_h
    n = tiles_x * tiles_y
    # slot-major order is tile order: tile t = k * world + r
    by_tile = gathered.permute(1, 0, 2, 3).reshape(n_pad * world, tile_h, tile_w, c)[:n]
    return by_tile.reshape(tiles_y, tiles_x, tile_h, tile_w, c).permute(0, 2, 1, 3, 4).reshape(height, width, c)


class FramePipeline:
    """Pipelined frame loop for N > 1 ranks: frame i's gather (RCCL, asynchronous) overlaps the trace of the
    following frames; rank 0 assembles a frame once its gather has completed.

    `trace` is a callable `trace(buf)` that enqueues the rank's tiles into `buf` ([n_pad, tile_h * tile_w, 4]
    int32) on the current stream -- or a list of such callables, one per LANE, together with `streams` (one
    torch.cuda.Stream per lane): consecutive frames then go to different HIP streams (each lane drives its own
    device context), so the serial tail of one frame's rays overlaps the next frames.  A rank's share of a
    sharded frame is small, and a ray is a serial chain of up to 101 dependent rounds: with one frame at a
    time the GPU idles through every frame's tail (tools/pipeline_probe.py: 1/8 of a 1080p frame takes 0.16 ms
    alone and 0.072 ms with three frames in flight).

    step() returns the most recent COMPLETED frame on rank 0 (None until the first one is ready, and on
    other ranks; with `assemble` it lives in a buffer that is reused after as many further steps as there are
    buffers); drain() completes what is in flight and returns the last frame."""

    def __init__(self, trace, width, height, tile_w, tile_h, rank, world, device, group=None, streams=None, assemble=None,
                 pack=None, gather=None, words=4):
        """assemble (optional, rank 0): one callable per lane, `assemble(gathered, out) -> out`, that un-permutes a gathered
        frame on the lane's stream (Render.assemble_tiles: one kernel, a few microseconds of host time); without it the
        generic torch expression assemble_frame() is used."""
        # gather (optional): one pair of callables per lane, (`gather(send, recv_on_root)`, `wait()`): the first enqueues this
        # rank's part of the frame-end gather behind the lane's stream (Gpu.gather_frame: RCCL behind the C ABI,
        # svo_gather_frame), the second makes the lane's stream wait for it (Gpu.gather_wait).  Without it the exchange is
        # torch.distributed's gather on the default process group (RCCL as well on the GPU box, gloo in the CPU tests)
        # words: int32 words per pixel slot of what `trace` produces: 4 = hit records (the default), 1 = RGBA8 colour (the
        # ranks shade their tiles and the frame that travels is the image, 4 bytes per ray; no packing then)
        self.words = words
        if words not in (1, 4) or (words == 1 and pack):
            raise ValueError("words must be 4 (records; optionally packed to 3 on the wire) or 1 (RGBA8, never packed)")
        self.gather = list(gather) if gather is not None else None
        self.traces = list(trace) if isinstance(trace, (list, tuple)) else [trace]
        self.assemble = list(assemble) if assemble is not None else None
        # pack: what goes over the links is the 12-byte wire form of the records.  True = the torch expression
        # pack_records(); a list = one callable per lane, `pack(records, wire) -> wire` (Render.pack_records)
        self.pack = pack
        self.streams = list(streams) if streams is not None else None
        if self.streams is not None and len(self.streams) != len(self.traces):
            raise ValueError("one stream per lane")
        self.rank, self.world, self.group = rank, world, group
        self.dims = (width, height, tile_w, tile_h)
        self.n_buf = max(2, len(self.traces))  # a lone lane is still double-buffered against its gather
        n_pad = padded_tile_count(width, height, tile_w, tile_h, world)
        self.local = [torch.zeros((n_pad, tile_h * tile_w, words), dtype=torch.int32, device=device) for _ in range(self.n_buf)]
        words = 3 if pack else words
        self.wire = [torch.zeros((n_pad, tile_h * tile_w, 3), dtype=torch.int32, device=device) if pack else None
                     for _ in range(self.n_buf)]
        self.gathered = [torch.empty((world, n_pad, tile_h * tile_w, words), dtype=torch.int32, device=device)
                         if rank == 0 else None for _ in range(self.n_buf)]
        self.frames = [torch.empty((height, width, self.words), dtype=torch.int32, device=device) if (rank == 0 and self.assemble) else None
                       for _ in range(self.n_buf)]
        self.recv = [list(g.unbind(0)) if g is not None else None for g in self.gathered]  # gather's output list, built once
        self.work = [None] * self.n_buf
        self.frame = None
        self.i = 0

    def _on_lane(self, b):
        """make the stream of the lane that owns buffer b the current one (torch.cuda.set_stream: 0.4 us; the
        torch.cuda.stream() context manager costs 6 us per use, which matters at 10 000 frames per second)"""
        if self.streams is not None:
            torch.cuda.set_stream(self.streams[b % len(self.traces)])

    def _finish(self, b):
        if self.work[b] is not None:
            if self.work[b] is True:
                self.gather[b % len(self.traces)][1]()  # C ABI: the lane's stream waits for the lane's gathers
            else:
                self.work[b].wait()  # the current stream waits for the collective
            self.work[b] = None
            if self.rank == 0:
                if self.assemble:
                    self.frame = self.assemble[b % len(self.traces)](self.gathered[b], self.frames[b])
                else:
                    self.frame = assemble_frame(self.gathered[b], *self.dims).contiguous()

    def step(self):
        b = self.i % self.n_buf
        self._on_lane(b)
        self._finish(b)      # buffer b is free again once the frame that used it has been gathered
        lane = b % len(self.traces)
        self.traces[lane](self.local[b])
        send = self.local[b]
        if self.pack:
            if self.pack is True:
                self.wire[b].copy_(pack_records(self.local[b]))
            else:
                self.pack[lane](self.local[b], self.wire[b])
            send = self.wire[b]
        if self.gather is not None:
            self.gather[lane][0](send, self.gathered[b] if self.rank == 0 else None)
            self.work[b] = True
        elif self.rank == 0:
            self.work[b] = dist.gather(send, self.recv[b], dst=0, group=self.group, async_op=True)
        else:
            self.work[b] = dist.gather(send, None, dst=0, group=self.group, async_op=True)
        self.i += 1
        return self.frame

    def drain(self):
        for k in range(self.n_buf):  # oldest frame first, so that self.frame ends up as the newest
            b = (self.i + k) % self.n_buf
            self._on_lane(b)
            self._finish(b)
        if self.streams is not None:
            for st in self.streams:
                st.synchronize()
            torch.cuda.set_stream(self.streams[0])  # lane 0 runs on the stream that was current at construction
        return self.frame
