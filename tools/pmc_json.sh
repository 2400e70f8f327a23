#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_json.sh <tag> [bench args]  ->  gpurun_out/<tag>/pmc.json
# Three separate rocprofv3 --pmc passes over the SAME command (bench.py, one step): SQ counters (VALU instructions, active lanes,
# busy cycles), FETCH_SIZE, WRITE_SIZE — they do not fit one pass (MI355X_MICROARCH.md "rocprofv3 PMC slots"); only --kernel-trace
# beside --pmc.  Counter collection serialises the dispatches, so the per-kernel figures are those of a kernel running alone.
# gfx950 corrections as the guide prescribes: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2 for STREAMING kernels (per-pixel records and queues
# read by consecutive lanes).  GATHER kernels (traversal, neighbour gathers, the global-gather candidate pick) issue 64-B requests that are counted at face value, so for
# them both figures are kept: hbm_bytes_per_launch = FETCH + WRITE (the estimate), hbm_bytes_per_launch_corrected = 2 x FETCH + WRITE (the upper bound; what rounds 2 - 5
# reported for every kernel).  FETCH_SIZE / WRITE_SIZE are in KiB.
tag=${1:-pmc}; shift; args="$@"; R=$PWD; mkdir -p gpurun_out/$tag
pass() {  # name, counters...
  name=$1; shift
  (cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$tag/$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-exact --no-other-reuse $args > $R/gpurun_out/$tag/$name.log 2>&1)
}
pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = collections.defaultdict(dict)
def load(name):
    f = glob.glob(f"gpurun_out/{tag}/{name}/*/*counter_collection.csv")
    if not f: return
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if not k.startswith("lm_k") or k.endswith("_inst"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in agg:
        n = len(disp[k]); out[k].setdefault("launches", n)
        for c, v in agg[k].items(): out[k][c + "_per_launch"] = v / n
    # kernel durations of this (serialised) pass
    t = glob.glob(f"gpurun_out/{tag}/{name}/*/*kernel_trace.csv")
    if t and name == "sq":
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(t[0])):
            k = r["Kernel_Name"]
            if k in agg: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items(): out[k]["alone_us"] = sum(v) / len(v)
for name in ("sq", "fetch", "write"): load(name)
import re
GATHER = re.compile(r"lm_k_(trace_|query_|path_tail|restir_trace_shade|restir_spatial|pick_primary(_fast|_rare)?$|shade_wave|refit_|kat_)")      # who gathers (64-B requests); the rest stream
for k, v in out.items():
    f, w = v.get("FETCH_SIZE_per_launch", 0.0) * 1024, v.get("WRITE_SIZE_per_launch", 0.0) * 1024
    v["fetch_bytes_raw_per_launch"] = f; v["write_bytes_per_launch"] = w; v["hbm_bytes_per_launch_corrected"] = 2 * f + w
    v["access"] = "gather" if GATHER.match(k) else "stream"
    v["hbm_bytes_per_launch"] = (f if v["access"] == "gather" else 2 * f) + w
    if v.get("SQ_INSTS_VALU_per_launch"):
        v["active_lanes_per_valu_inst"] = v.get("SQ_THREAD_CYCLES_VALU_per_launch", 0.0) / v["SQ_INSTS_VALU_per_launch"]
    if v.get("SQ_BUSY_CYCLES_per_launch"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs of the chip; SQ_BUSY_CYCLES is summed over the shader engines' SQs
        v["valu_quadcycles_per_launch"] = v.get("SQ_ACTIVE_INST_VALU_per_launch", 0.0)
import hashlib, os
def ksid():
    h = hashlib.sha256(); d = "lumenrenderer_amd/csrc"
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")): h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]
json.dump({"kernel_source_id": ksid(), "note": "per launch; counter passes serialise the dispatches (alone times); hbm_bytes_per_launch = FETCH_SIZE x (2 for streaming kernels, 1 for gather kernels: field access) + WRITE_SIZE; hbm_bytes_per_launch_corrected = 2 x FETCH_SIZE + WRITE_SIZE for every kernel (upper bound)",
           "kernels": out}, open(f"gpurun_out/{tag}/pmc.json", "w"), indent=1, sort_keys=True)
tot = est = 0.0
for k in sorted(out, key=lambda k: -out[k].get("alone_us", 0) * out[k]["launches"]):
    v = out[k]; tot += v["hbm_bytes_per_launch_corrected"] * v["launches"]; est += v["hbm_bytes_per_launch"] * v["launches"]
    print(f'{k:32s} x{v["launches"]:3d} alone {v.get("alone_us", 0):8.1f} us  valu insts {v.get("SQ_INSTS_VALU_per_launch", 0):12.4g}  lanes/inst {v.get("active_lanes_per_valu_inst", 0):5.1f}  {v["access"]:6s} hbm {v["hbm_bytes_per_launch"] / 1e6:9.1f} MB (upper {v["hbm_bytes_per_launch_corrected"] / 1e6:9.1f})')
print("sum hbm bytes over the run: estimate", est / 1e9, "GB, upper bound (2 x FETCH everywhere)", tot / 1e9, "GB")
PY
rm -rf gpurun_out/$tag/sq gpurun_out/$tag/fetch gpurun_out/$tag/write
