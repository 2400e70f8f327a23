"""Closed tessellated surfaces and edge / vertex-aimed rays for the watertightness property of the hit rule (tests only).

What the reference guarantees (its geometry queries go through OptiX: WaveFrontShaders.cu:63-76,128-140,197-210 on a GAS built by
OptixWrapper.cpp:46-78): a ray cannot pass BETWEEN two triangles that share an edge or a vertex.  The generators below build meshes
whose neighbouring triangles share their vertices bit for bit (indexed geometry) and rays that start inside and aim at points ON
shared edges and AT shared vertices, moved by a few units in the last place, so that every ray must report a hit."""
import numpy as np


def icosphere(level):
    """Indexed unit icosphere: 20 * 4**level triangles, vertices shared through the index buffer."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        mid = {}
        nf = []
        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p)); mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float64), np.array(f, np.uint32)


def quad_grid(nu, nv, origin, eu, ev):
    """A planar grid of nu x nv quads (two triangles each, alternating diagonals), vertices shared."""
    u, w = np.meshgrid(np.arange(nu + 1), np.arange(nv + 1), indexing="ij")
    pos = np.asarray(origin, np.float64) + u[..., None] * np.asarray(eu, np.float64) + w[..., None] * np.asarray(ev, np.float64)
    idx = lambda i, j: i * (nv + 1) + j
    f = []
    for i in range(nu):
        for j in range(nv):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            f += [(a, b, c), (a, c, d)] if (i + j) & 1 else [(a, b, d), (b, c, d)]
    return pos.reshape(-1, 3), np.array(f, np.uint32)


def mesh_scene(pos, faces, transform=None):
    """SceneDescription with one indexed primitive (+ the small emissive quad random_soup-style scenes need to render)."""
    from lumenrenderer_amd.scenes import SceneDescription, interleave
    d = SceneDescription()
    m = d.add_material(diffuse_color=(0.7, 0.7, 0.7, 1), metallic_factor=0.0, roughness_factor=0.8)
    pos = np.asarray(pos, np.float32)
    n = np.tile(np.float32([0, 1, 0]), (len(pos), 1)); tg = np.tile(np.float32([1, 0, 0, 1]), (len(pos), 1))
    uv = np.zeros((len(pos), 2), np.float32)
    d.add_instance(d.add_mesh([d.add_primitive(interleave(pos, uv, n, tg), np.asarray(faces, np.uint32).ravel(), m)]), transform)
    return d


def _ulp_nudge(x, rng, k):
    """x (float32) moved by an integer number of units in the last place in [-k, k]."""
    bits = x.view(np.int32).copy()
    step = rng.integers(-k, k + 1, x.shape).astype(np.int32)
    return np.where(np.isfinite(x) & (x != 0), (bits + np.where(bits >= 0, step, -step)).view(np.float32), x)


def seam_rays(world_tris, origin, n, seed, ulps=4, vertex_share=0.25):
    """n rays from `origin` (array [3] or [n, 3]) aimed at points on shared edges (a + s (b - a), evaluated in float64 and rounded
    once) and at vertices of `world_tris` ([T, 3, 3] float32 as the renderer holds them), each direction component then moved by up
    to `ulps` units in the last place.  Directions are NOT normalised (the hit rule must not depend on it) for half of the rays."""
    rng = np.random.default_rng(seed)
    wt = np.asarray(world_tris, np.float64)
    tri = rng.integers(0, len(wt), n)
    e = rng.integers(0, 3, n)
    a, b = wt[tri, e], wt[tri, (e + 1) % 3]
    s = rng.uniform(0.0, 1.0, (n, 1))
    s[rng.uniform(size=n) < 0.1] = 0.5                                   # exact midpoints
    target = a + s * (b - a)
    at_vertex = rng.uniform(size=n) < vertex_share
    target[at_vertex] = a[at_vertex]
    org = np.broadcast_to(np.asarray(origin, np.float32), (n, 3)).astype(np.float32)
    d = (target - org.astype(np.float64))
    unit = rng.uniform(size=n) < 0.5
    d[unit] /= np.linalg.norm(d[unit], axis=1, keepdims=True)
    d = _ulp_nudge(d.astype(np.float32), rng, ulps)
    return org, d, target
