out=gpurun_out/rv
mkdir -p $out
export TMPDIR=/tmp
DASP_RV_NOISE=generated bash scripts/reverb_traffic.sh $out/hbm_traffic_secondary.json 2>&1 | tail -2
( cd /tmp && DASP_RV_NOISE=generated rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/reverb_time.py 128 2 262144 > /dev/null 2>> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/reverb_kernel_stats.csv; rm -rf $out/rprof
cut -d, -f1-4 $out/reverb_kernel_stats.csv | grep dasp | sed 's/(float[^"]*"/"/' | head -16
python - <<'PY'
import json
t=json.load(open("gpurun_out/rv/hbm_traffic_secondary.json"))
for k,v in t["kernels"].items(): print(k, round(v["hbm_bytes"]/1e9,3))
print({k:v for k,v in t.items() if k not in ("kernels","note")})
PY
