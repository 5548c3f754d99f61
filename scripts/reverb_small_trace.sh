#!/bin/bash
# per-kernel durations of the reverb step (in-kernel noise) at the reference's training sizes: rocprofv3 --kernel-trace --stats
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
for shp in "16 1 131072" "8 2 131072"; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/reverb_time.py $shp > $GRAFT_REPO_ROOT/$out/rv_small.out 2> $GRAFT_REPO_ROOT/$out/rprof.err )
echo "== $shp"; tail -1 $out/rv_small.out | cut -c1-400
python - <<PY
import csv, glob
f = glob.glob("$out/rprof/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "dasp" in r["Name"]]
tot = 0.0
for r in rows:
    per_step = float(r["TotalDurationNs"]) / 80.0 / 1e3
    tot += per_step
    print(f"  {r['Name'][:70]:70s} calls/step {int(r['Calls']) / 80:.1f}  avg {float(r['AverageNs']) / 1e3:7.1f} us  per step {per_step:7.1f} us")
print(f"  sum per step {tot:.1f} us")
PY
rm -rf $out/rprof
done
