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


def assemble_frame(gathered, width, height, tile_w, tile_h):
    """[world, n_pad, tile_h*tile_w, 4] (rank r, slot k holds tile r + k*world) -> [H, W, 4]."""
    world, n_pad = gathered.shape[0], gathered.shape[1]
    tiles_x, tiles_y = width // tile_w, height // tile_h
    n = tiles_x * tiles_y
    # slot-major order is tile order: tile t = k * world + r
    by_tile = gathered.permute(1, 0, 2, 3).reshape(n_pad * world, tile_h, tile_w, 4)[:n]
    return by_tile.reshape(tiles_y, tiles_x, tile_h, tile_w, 4).permute(0, 2, 1, 3, 4).reshape(height, width, 4)
