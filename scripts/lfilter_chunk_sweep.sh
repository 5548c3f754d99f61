#!/bin/bash
# lfilter_via_fsm (K > 3) forward + backward time against the chunk length, per kernel (rocprofv3) for the default chunk
for c in 128 256 512 1024 2048; do echo "chunk $c"; DASP_LFILTER_CHUNK=$c python scripts/lfilter_long_time.py 2>/dev/null | head -2; done > gpurun_out/lfilter_chunk_sweep.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/lf_prof -o lf -- python $GRAFT_REPO_ROOT/scripts/lfilter_long_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls gpurun_out/lf_prof/*/*kernel_stats.csv gpurun_out/lf_prof/*kernel_stats.csv 2>/dev/null | head -1); python - "$f" >> gpurun_out/lfilter_chunk_sweep.log <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us', r['Percentage'])
PY
