"""Watertightness of the hit rule through the C ABI's ray-query seam (lumen_mi_query_closest / _any = OptixWrapper::TraceRays,
OptixWrapper.h:58-81).  The reference's queries run on OptiX triangle GASes (OptixWrapper.cpp:46-131, WaveFrontShaders.cu:63-76),
where a ray cannot slip between two triangles that share an edge; this suite holds the HIP traversal to the same property: more than
ten million rays aimed at shared edges and vertices of closed / gap-free meshes, zero escapes, and the reported hit lies on a triangle
that touches the target with the float64 distance."""
import numpy as np
import pytest

from helpers import product_from
from watertight import icosphere, quad_grid, mesh_scene, seam_rays

pytestmark = pytest.mark.gpu


def _rot(axis, angle):
    axis = np.asarray(axis, np.float64); axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def _xf(scale, rot, translate):
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = rot @ np.diag(np.broadcast_to(np.float64(scale), 3))
    m[:3, 3] = translate
    return m.astype(np.float32)


SPHERES = [("unit", _xf(1.0, np.eye(3), (0, 0, 0))),
           ("0.008 at an offset", _xf(0.008, _rot((1, 2, 3), 0.7), (0.31, 0.17, -0.23))),
           ("100 far from the origin", _xf(100.0, _rot((-1, 0.3, 2), 2.1), (1000.0, -2000.0, 500.0))),
           ("3 x 0.1 x 0.008", _xf((3.0, 0.1, 0.008), _rot((0.2, 1, -0.4), 1.3), (-7.3, 2.9, 13.7)))]


def _check(r, org, d, target, wt, what, chunk=1 << 21):
    """every ray hits (closest and any); the closest hit lies on a triangle touching the target point at the float64 distance"""
    n = len(org)
    escapes_c = escapes_a = 0
    worst = 0.0
    vkey = {}
    for a in range(0, n, chunk):
        o_, d_, tg = org[a:a + chunk], d[a:a + chunk], target[a:a + chunk]
        ip, uvt = r.QueryClosest(o_, d_, 1e-6, 1e30)
        occ = r.QueryAny(o_, d_, np.full(len(o_), 1e30, np.float32), tmin=1e-6)
        hit = uvt[:, 2] > 0
        escapes_c += int((~hit).sum()); escapes_a += int((occ == 0).sum())
        # distance along the ray in float64 to the target point against the reported t
        dn = np.linalg.norm(d_.astype(np.float64), axis=1)
        want = np.linalg.norm(tg - o_.astype(np.float64), axis=1)
        got = uvt[:, 2].astype(np.float64) * dn
        rel = np.abs(got - want)[hit] / want[hit]
        worst = max(worst, float(rel.max()) if hit.any() else 0.0)
        # the triangle reported touches the target: one of its vertices is an end of the target's edge or the point lies in its plane within rounding
        tri = wt[ip[hit, 1].astype(np.int64)].astype(np.float64)
        nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        size = np.abs(tri).max(axis=(1, 2))
        off = np.abs(np.einsum("ij,ij->i", tg[hit] - tri[:, 0], nrm))
        edge = np.linalg.norm(tri[:, 1] - tri[:, 0], axis=1)
        assert (off <= 0.05 * edge + 1e-5 * size).all(), (what, float((off / edge).max()))
    assert escapes_c == 0 and escapes_a == 0, f"{what}: {escapes_c} closest-hit and {escapes_a} any-hit rays of {n} escaped through a shared edge / vertex"
    assert worst < 2e-4, (what, worst)


@pytest.mark.parametrize("case", range(len(SPHERES)))
def test_no_ray_leaves_a_closed_icosphere_through_a_shared_edge_or_vertex(case):
    name, xf = SPHERES[case]
    pos, faces = icosphere(5)                                            # 20 480 triangles
    d = mesh_scene(pos, faces, xf)
    r = product_from(d, 16, 16, 2)
    wt = r.GetWorldTriangles().reshape(-1, 3, 3)[: len(faces)]
    n = 2_600_000
    rng = np.random.default_rng(100 + case)
    inside = rng.normal(size=(n, 3)); inside *= (0.6 * rng.uniform(size=(n, 1)) ** (1 / 3)) / np.linalg.norm(inside, axis=1, keepdims=True)
    inside[: n // 4] = 0.0                                               # a quarter from the centre itself
    m = xf.astype(np.float64)
    origin = (inside @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
    org, dr, target = seam_rays(wt, origin, n, seed=200 + case)
    _check(r, org, dr, target, wt, name)
    r.close()


@pytest.mark.parametrize("case", range(3))
def test_no_ray_passes_through_the_seams_of_a_coplanar_quad_grid(case):
    """coplanar quads at awkward scales (0.008 x 0.1: the stand-in atrium's instance scales), axis-aligned, tilted, and far from the origin"""
    eu, ev, org0 = [((0.008, 0, 0), (0, 0, 0.1), (13.7, 2.9, -7.3)),
                    ((0.008 * 0.6, 0.008 * 0.8, 0), (0, 0, 0.1), (-0.41, 0.77, 0.13)),
                    ((0.3, 0.1, 0.7), (-0.7, 0.2, 0.3 - 0.2 / 7), (900.0, 1200.0, -400.0))][case]
    nu, nv = 96, 64
    pos, faces = quad_grid(nu, nv, org0, eu, ev)
    d = mesh_scene(pos, faces)
    r = product_from(d, 16, 16, 2)
    wt_all = r.GetWorldTriangles().reshape(-1, 3, 3)[: len(faces)]
    # interior triangles only: every edge and vertex they own is shared with a neighbour
    gi, gj = np.divmod(faces.astype(np.int64), nv + 1)
    interior = ((gi > 0) & (gi < nu) & (gj > 0) & (gj < nv)).all(axis=1)
    wt = wt_all[interior]
    n = 1_000_000
    eu_, ev_ = np.float64(eu), np.float64(ev)
    nrm = np.cross(eu_, ev_); nrm /= np.linalg.norm(nrm)
    centre = np.float64(org0) + 0.5 * nu * eu_ + 0.5 * nv * ev_
    span = max(np.linalg.norm(nu * eu_), np.linalg.norm(nv * ev_))
    rng = np.random.default_rng(300 + case)
    origin = (centre + nrm * span * rng.uniform(0.2, 1.5, (n, 1)) + rng.uniform(-0.3, 0.3, (n, 1)) * nu * eu_ + rng.uniform(-0.3, 0.3, (n, 1)) * nv * ev_).astype(np.float32)
    org, dr, target = seam_rays(wt, origin, n, seed=400 + case)
    n_escape = 0
    for a in range(0, n, 1 << 20):
        ip, uvt = r.QueryClosest(org[a:a + (1 << 20)], dr[a:a + (1 << 20)], 1e-6, 1e30)
        occ = r.QueryAny(org[a:a + (1 << 20)], dr[a:a + (1 << 20)], np.full(len(org[a:a + (1 << 20)]), 1e30, np.float32), tmin=1e-6)
        n_escape += int((uvt[:, 2] <= 0).sum()) + int((occ == 0).sum())
    assert n_escape == 0, f"{n_escape} rays passed through a seam of the grid"
    r.close()
