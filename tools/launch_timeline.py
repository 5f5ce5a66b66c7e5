import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None; prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:60]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{name:60s} grid {r.get('Grid_Size','?'):>9s} dur {(e-s)/1e3:9.1f} us gap {gap:8.1f} us")
    prev_end = e
