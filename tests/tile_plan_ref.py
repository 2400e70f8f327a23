"""The tile plan restated in plain Python — test infrastructure: the checker of the C++ plan behind lumen_mi_group_plan / lumen_mi_group_seams
(csrc/group.cpp), which is the one implementation the product uses (lumenrenderer_amd/tiles.py calls into it).  Rounds 1 - 5 shipped this text as the product."""

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


def _intersect(a, b):
    x0, y0, x1, y1 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    return (x0, y0, x1, y1) if x0 < x1 and y0 < y1 else None


def halo_plan(rank, n, width, height):
    """What rank exchanges after every frame so that temporal reuse in its halo ring sees the owners' reservoirs: a list of
    (peer, send_rect, recv_rect) in global pixels — send = my tile inside the peer's window, recv = the peer's tile inside my
    window (either may be None).  The halo ring of a window is the disjoint union of the recv rectangles."""
    mine = tile_rect(rank, n, width, height)
    my_window = window_rect(mine, width, height)
    plan = []
    for peer in range(n):
        if peer == rank:
            continue
        theirs = tile_rect(peer, n, width, height)
        send = _intersect(mine, window_rect(theirs, width, height))
        recv = _intersect(theirs, my_window)
        if send or recv:
            plan.append((peer, send, recv))
    return plan


