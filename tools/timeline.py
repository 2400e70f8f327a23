"""Print a per-stream timeline of the middle of a rocprofv3 kernel trace: python tools/timeline.py <kernel_trace.csv> [t0_us] [t1_us]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("lm_k") and not r["Kernel_Name"].endswith("_inst")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prim = [i for i, r in enumerate(rows) if r["Kernel_Name"] == "lm_k_primary"]
s = prim[-3]                                   # start of the third-last TraceFrame: steady state of the pipeline
t0 = int(rows[s]["Start_Timestamp"])
end = prim[-1]
streams = sorted({r["Queue_Id"] for r in rows})
for r in rows[s:end]:
    a = (int(r["Start_Timestamp"]) - t0) / 1e3; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    col = streams.index(r["Queue_Id"])
    print(f"{a:9.1f} {d:8.1f}  " + "                    " * col + r["Kernel_Name"][5:])
