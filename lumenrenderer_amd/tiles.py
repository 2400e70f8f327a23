"""Tile sharding of one frame over the GPUs of a node (new functionality; the reference is single-GPU).

One process per GPU.  The scene, BVH and light list are replicated; the image is cut into a cols x rows grid; every rank
renders a *window* = its tile grown by a halo of 60 px (clamped to the image), because ReSTIR's two spatial passes read
neighbours within +-30 px each (ReSTIRData.h:46-56).  RNG streams, Halton indices and light-bag tiles are functions of the
GLOBAL pixel coordinate, so a tile pixel whose whole 60-px neighbourhood lies inside the window gets exactly the
single-GPU value for a first frame; across blended frames the temporal history of the outer halo ring is window-local
(DESIGN.md §Multi-GPU states the seam behaviour).  The only collective is one gather of the final fp32 radiance tiles
(RCCL over xGMI when the backend is nccl; gloo on CPU in the tests).
"""
import math

HALO = 60


def grid_for(n, width, height):
    """cols x rows with cols*rows == n whose LARGEST rank window (tile + halo, clipped to the image) is smallest: the frame time
    of the slowest rank is what the gather waits for."""
    best = None
    for cols in range(1, n + 1):
        if n % cols:
            continue
        rows = n // cols
        worst = 0
        for cy in range(rows):
            for cx in range(cols):
                x0, x1 = (width * cx) // cols, (width * (cx + 1)) // cols
                y0, y1 = (height * cy) // rows, (height * (cy + 1)) // rows
                wx0, wy0, wx1, wy1 = max(0, x0 - HALO), max(0, y0 - HALO), min(width, x1 + HALO), min(height, y1 + HALO)
                worst = max(worst, (wx1 - wx0) * (wy1 - wy0))
        if best is None or worst < best[0]:
            best = (worst, cols, rows)
    return best[1], best[2]


def tile_rect(rank, n, width, height):
    cols, rows = grid_for(n, width, height)
    cx, cy = rank % cols, rank // cols
    x0, x1 = (width * cx) // cols, (width * (cx + 1)) // cols
    y0, y1 = (height * cy) // rows, (height * (cy + 1)) // rows
    return x0, y0, x1, y1


def window_rect(tile, width, height, halo=HALO):
    x0, y0, x1, y1 = tile
    return max(0, x0 - halo), max(0, y0 - halo), min(width, x1 + halo), min(height, y1 + halo)


def max_tile_shape(n, width, height):
    rects = [tile_rect(r, n, width, height) for r in range(n)]
    return max(r[3] - r[1] for r in rects), max(r[2] - r[0] for r in rects)


def gather_tiles(local_tile, rank, world, width, height, dist, dst=0):
    """local_tile: torch tensor [th, tw, 4] (this rank's tile, halo removed).  Returns the assembled [H, W, 4] image on
    ``dst`` (None elsewhere).  Tiles are padded to a common shape so that one gather moves them."""
    import torch
    mh, mw = max_tile_shape(world, width, height)
    pad = torch.zeros((mh, mw, 4), dtype=local_tile.dtype, device=local_tile.device)
    pad[: local_tile.shape[0], : local_tile.shape[1]] = local_tile
    if world == 1:
        parts = [pad]
    else:
        parts = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, parts, dst=dst)
    if rank != dst:
        return None
    img = torch.empty((height, width, 4), dtype=local_tile.dtype, device=local_tile.device)
    for r in range(world):
        x0, y0, x1, y1 = tile_rect(r, world, width, height)
        img[y0:y1, x0:x1] = parts[r][: y1 - y0, : x1 - x0]
    return img
