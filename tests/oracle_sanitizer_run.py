"""Drives the oracle through every entry point the parity tests use; run by tests/test_cpu_host.py in a subprocess against an
AddressSanitizer + UBSan build of oracle/lumen_oracle.cpp (LUMEN_ORACLE_SO).  Test infrastructure only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import cornell, oracle_from  # noqa: E402


def main():
    d = cornell()
    o = oracle_from(d, 40, 28, 4, blend=True)
    base = np.array(d.instances[0]["transform"], np.float32).reshape(4, 4)
    for f in range(4):
        if f == 1:
            o.set_camera((0.05, 1.0, 3.2), (1, 0, 0), (0, 1, 0), (0, 0, -1), 70.0)
        if f == 2:
            m = base.copy(); m[1, 3] += 0.01
            o.set_instance_transform(0, m)
            o.set_instance_emissiveness(0, 0, (0, 0, 0), 2.0)
        if f == 3:
            o.set_window(3, 5, 37, 21)
        assert o.trace_frame() == 0
    rad = o.radiance()
    assert np.isfinite(rad).all()
    o.channel(0); o.channel(1); o.output_pixels(); o.stats(24); o.gbuffer(); o.denoiser_inputs(); o.lights(); o.world_triangles()
    rng = np.random.default_rng(3)
    org = rng.uniform(-1, 1, (64, 3)).astype(np.float32) + np.float32([0, 1, 3])
    dr = rng.normal(size=(64, 3)).astype(np.float32); dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    for use_bvh in (True, False):
        o.trace_closest(org, dr, use_bvh=use_bvh)
        o.trace_any(org, dr, np.full(64, 10.0, np.float32), use_bvh=use_bvh)
    o.close()
    # a one-pixel image, depth 1, and the deepest path the counters allow
    for w, h, depth in ((1, 1, 1), (9, 7, 16)):
        o = oracle_from(d, w, h, depth)
        assert o.trace_frame() == 0
        o.close()
    print("oracle sanitizer run ok")


if __name__ == "__main__":
    main()
