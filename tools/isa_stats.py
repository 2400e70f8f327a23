#!/usr/bin/env python3
"""Static ISA statistics per kernel of kernels.hip (gfx950): VGPRs, scratch, instruction count, IEEE-division / square-root / reciprocal
sequences.  Runs anywhere hipcc is (no GPU needed): `python3 tools/isa_stats.py [EXTRA flags] > profiles/rNN_isa_stats.txt`."""
import re, subprocess, sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "lumenrenderer_amd", "csrc")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-DLM_INSTRUMENT=0", "--cuda-device-only", "-S",
           os.path.join(csrc, "kernels.hip"), "-o", out] + sys.argv[1:]
    subprocess.check_call(cmd)
    text = open(out).read()
kern = {}
cur = None
for line in text.splitlines():
    m = re.match(r"^(lm_k_\w+):", line)
    if m:
        cur = m.group(1); kern[cur] = {"insts": 0, "div": 0, "sqrt": 0, "rcp": 0, "rsq": 0, "trans": 0, "scratch": 0}
        continue
    if cur is None:
        continue
    if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
        cur = None; continue
    t = line.strip()
    if not t or t.startswith((";", ".", "s_nop")) or t.endswith(":"):
        continue
    op = t.split()[0]
    k = kern[cur]
    k["insts"] += 1
    if op.startswith("v_div_fixup"): k["div"] += 1
    if op.startswith("v_sqrt_f32"): k["sqrt"] += 1
    if op.startswith("v_rcp_f32"): k["rcp"] += 1
    if op.startswith("v_rsq_f32"): k["rsq"] += 1
    if op.startswith(("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")): k["trans"] += 1
    if op.startswith("scratch_"): k["scratch"] += 1
for m in re.finditer(r"\.amdhsa_kernel (lm_k_\w+)(.*?)\.end_amdhsa_kernel", text, re.S):
    name, body = m.group(1), m.group(2)
    if name in kern:
        v = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body); a = re.search(r"\.amdhsa_accum_offset (\d+)", body)
        p = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body); l = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body)
        kern[name].update(vgpr_total=int(v.group(1)) if v else -1, arch_vgpr=int(a.group(1)) if a else -1, scratch_bytes=int(p.group(1)) if p else -1, lds=int(l.group(1)) if l else -1)
print(f"{'kernel':34s} {'insts':>6s} {'arch_vgpr':>9s} {'vgpr+agpr':>9s} {'scratchB':>8s} {'scr_ins':>7s} {'lds':>6s} {'ieee_div':>8s} {'v_sqrt':>6s} {'v_rcp':>6s} {'v_rsq':>6s}")
for name in sorted(kern):
    k = kern[name]
    print(f"{name:34s} {k['insts']:6d} {k.get('arch_vgpr',-1):9d} {k.get('vgpr_total',-1):9d} {k.get('scratch_bytes',-1):8d} {k['scratch']:7d} {k.get('lds',-1):6d} {k['div']:8d} {k['sqrt']:6d} {k['rcp']:6d} {k['rsq']:6d}")
