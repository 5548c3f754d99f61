#!/bin/bash
# kernel time line (start/end, gaps) of one device draw at the reverb's noise shape for BS items: rocprofv3 --kernel-trace
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
for b in ${BS:-8}; do
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/rtrace -o p -- python $GRAFT_REPO_ROOT/scripts/mtrand_time.py $b > /dev/null 2> $GRAFT_REPO_ROOT/$out/rtrace.err )
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/rtrace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob("$out/rtrace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
rows.sort()
last = [i for i, r in enumerate(rows) if "mt_generate" in r[2]][-1]
first = max(i for i in range(last) if "mt_seed" in rows[i][2])
t0 = rows[first][0]
for s, e, n in rows[max(0, first - 2):last + 4]:
    print(f"b$b  start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  {n}")
PY
rm -rf $out/rtrace
done
