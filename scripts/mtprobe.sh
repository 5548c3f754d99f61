#!/bin/bash
# kernel times of probe builds of csrc/mtrand.hip (tools/mtprobe/*.so, built by scripts/mtprobe_build.sh): rocprofv3 kernel stats per build
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
for so in tools/mtprobe/*.so; do
  for b in ${MTPROBE_BS:-8}; do
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/mtrand_raw.py $GRAFT_REPO_ROOT/$so $b > /dev/null 2> $GRAFT_REPO_ROOT/$out/rprof.err )
    echo "$so b$b: $(grep mt_generate $(find $out/rprof -name '*kernel_stats.csv' | head -1) | awk -F, '{print "generate avg", $(NF-4), "min", $(NF-2), "max", $(NF-1)}')  $(grep mt_jump $(find $out/rprof -name '*kernel_stats.csv' | head -1) | awk -F, '{print "jump avg", $(NF-4), "max", $(NF-1)}')"
    rm -rf $out/rprof
  done
done
