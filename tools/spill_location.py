#!/usr/bin/env python3
"""Where do a kernel's scratch (spill) instructions sit?  For every kernel of kernels.hip that has any: each scratch instruction with the loop nesting
depth of its basic block, from the compiler's own block annotations in the -S output ("in Loop: Header=... Depth=N").  No GPU needed.
  python3 tools/spill_location.py > profiles/rNN_spill_location.txt"""
import re, subprocess, sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "lumenrenderer_amd", "csrc")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-DLM_INSTRUMENT=0", "--cuda-device-only", "-S",
                           os.path.join(csrc, "kernels.hip"), "-o", out] + sys.argv[1:], stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
cur, depth, maxdepth, rows, insts_at = None, 0, {}, {}, {}
for line in lines:
    m = re.match(r"^(lm_k_\w+):", line)
    if m:
        cur, depth = m.group(1), 0; rows[cur] = []; maxdepth[cur] = 0; insts_at[cur] = {}
        continue
    if cur is None:
        continue
    if line.startswith(".Lfunc_end"):
        cur = None; continue
    m = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?$", line)
    if m:
        d = re.search(r"Depth=(\d+)", line)
        depth = int(d.group(1)) if d else 0
        maxdepth[cur] = max(maxdepth[cur], depth)
        continue
    t = line.strip()
    if not t or t.startswith((";", ".")):
        continue
    insts_at[cur][depth] = insts_at[cur].get(depth, 0) + 1
    if t.startswith("scratch_"):
        rows[cur].append((depth, t))
for k in sorted(rows):
    if not rows[k]:
        continue
    print(f"{k}: loop nest depth {maxdepth[k]} (instructions per depth: {dict(sorted(insts_at[k].items()))})")
    for d, t in rows[k]:
        print(f"    depth {d}   {t}")
