import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if "sos_" in name:
        key = (name.split("(")[0].replace("void dasp::", "")[:40], r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v = sorted(v[5:])
    print(k, "n", len(v), "median us", round(v[len(v) // 2] / 1e3, 2))
