#!/bin/bash
# Roofline figures of the multi-resolution STFT loss kernels (csrc/stftloss.hip) at (16, 2, 131072), three default resolutions:
#   pass 1  rocprofv3 --kernel-trace --stats          -> average kernel durations (<out>/mrstft_kernel_stats.csv)
#   pass 2  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES (with --kernel-trace only)
#   pass 3/4  --pmc FETCH_SIZE / --pmc WRITE_SIZE      (separate passes: they do not fit one; FETCH_SIZE doubled per MI355X_MICROARCH.md)
# and writes <out>/mrstft_roofline.json: per kernel the VALU instructions per launch, the fraction of the fp32 vector peak they amount to at
# the measured duration (peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T lane-operations/s = 157.3 TFLOP/s counting an FMA as two), and
# the HBM bytes against 8 TB/s. usage (GPU box): scripts/mrstft_roofline.sh profiles/r03
out=${1:-gpurun_out/mrstft}
mkdir -p "$out" gpurun_out/pmc_stft; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_stft/*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_stft/stats -o p -- python scripts/loss_time.py > gpurun_out/pmc_stft/stats.log 2>&1 || echo "stats pass failed"
for grp in "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '+')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_stft/$tag -o p -- python scripts/loss_time.py > gpurun_out/pmc_stft/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, shutil, sys
out = sys.argv[1]
short = lambda n: n.split("dasp::")[1].split("(")[0] if "dasp::" in n else None
dur = {}
for path in glob.glob("gpurun_out/pmc_stft/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(path, out + "/mrstft_kernel_stats.csv")
    for r in csv.DictReader(open(path)):
        k = short(r["Name"])
        if k:
            dur[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/pmc_stft/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9
res = {"note": "scripts/mrstft_roofline.sh: rocprofv3 passes over scripts/loss_time.py, (16, 2, 131072), resolutions (1024,120,600) (2048,240,1200) "
               "(512,50,240); counters per launch (second half of the launches), durations from the --stats pass. valu_frac_of_fp32_peak = "
               "SQ_INSTS_VALU x 64 lanes / duration / (256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz): the share of the chip's vector issue slots the "
               "kernel's VALU instructions occupy (an upper bound on its flop fraction: not every VALU instruction is an FMA). hbm_bytes = 2 x "
               "FETCH_SIZE + WRITE_SIZE (KB units; gfx950 correction of MI355X_MICROARCH.md).", "shape": [16, 2, 131072], "kernels": {}}
tot_t = tot_v = tot_b = 0.0
for k in sorted(dur):
    c = {n: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for n, v in cnt.get(k, {}).items()}
    t = dur[k]["avg_us"] * 1e-6
    e = dict(dur[k])
    if "SQ_INSTS_VALU" in c:
        e["valu_insts"] = c["SQ_INSTS_VALU"]
        e["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1) if c.get("SQ_WAVES") else None
        e["valu_frac_of_fp32_peak"] = round(c["SQ_INSTS_VALU"] * 64 / t / PEAK_LANE_OPS, 4)
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        e["hbm_bytes"] = int(2 * c.get("FETCH_SIZE", 0) * 1024 + c.get("WRITE_SIZE", 0) * 1024)
        e["hbm_frac_of_8TBps"] = round(e["hbm_bytes"] / t / 8e12, 4)
    res["kernels"][k] = e
    if "stft" in k and "twiddle" not in k:
        per_step = dur[k]["calls"]
        tot_t += dur[k]["total_ms"]; tot_v += e.get("valu_insts", 0) * per_step; tot_b += e.get("hbm_bytes", 0) * per_step
steps = 71.0       # loss_time.py: 20 + 50 timed + 1 final forward
res["per_step"] = {"gpu_ms": round(tot_t / steps, 4)}
json.dump(res, open(out + "/mrstft_roofline.json", "w"), indent=1)
print(json.dumps(res["kernels"], indent=0)[:3000])
PY
rm -rf gpurun_out/pmc_stft
