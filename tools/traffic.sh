#!/bin/bash
# HBM traffic per kernel from PMC counters (GPU box): FETCH_SIZE and WRITE_SIZE in two separate rocprofv3 --pmc passes
# (they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"); output gpurun_out/<tag>/traffic.json
tag=${1:-traffic}; R=$PWD; mkdir -p gpurun_out/$tag
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/$tag/$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$tag/$c.log 2>&1
  cd $R
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag=sys.argv[1]; out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"gpurun_out/{tag}/{c}/*/*counter_collection.csv")[0]
    agg=collections.defaultdict(float); calls=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not k.startswith("lm_k") or k.endswith("_inst") or r["Counter_Name"]!=c: continue
        agg[k]+=float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    for k in agg: out.setdefault(k,{})[c]={"sum_kb":agg[k],"launches":len(calls[k])}
res={}
for k,v in out.items():
    n=max(v.get("FETCH_SIZE",{}).get("launches",1),1)
    fetch=v.get("FETCH_SIZE",{}).get("sum_kb",0.0)*1024/n; write=v.get("WRITE_SIZE",{}).get("sum_kb",0.0)*1024/max(v.get("WRITE_SIZE",{}).get("launches",1),1)
    # gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md "HBM"); gather widths are uncalibrated
    res[k]={"launches":n,"fetch_bytes_raw_per_launch":fetch,"write_bytes_per_launch":write,"hbm_bytes_per_launch_corrected":2*fetch+write}
json.dump(res, open(f"gpurun_out/{tag}/traffic.json","w"), indent=1, sort_keys=True)
for k in sorted(res,key=lambda k:-res[k]["hbm_bytes_per_launch_corrected"]*res[k]["launches"]): print(k.ljust(32), res[k]["launches"], f'{res[k]["hbm_bytes_per_launch_corrected"]/1e6:10.1f} MB/launch')
PY
rm -rf gpurun_out/$tag/FETCH_SIZE gpurun_out/$tag/WRITE_SIZE
