#!/bin/bash
# same-box A/B of the HOST side of the small-batch steps: the round-5 tree (tools/r05tree: commit db1df0e, built) against this tree
out=gpurun_out/r06/eager_host_ab.log; mkdir -p gpurun_out/r06; : > $out
for i in 1 2 3; do
  PYTHONPATH=$GRAFT_REPO_ROOT/tools/r05tree python scripts/eager_host_ab.py r05 2>/dev/null | tail -1 >> $out
  PYTHONPATH=$GRAFT_REPO_ROOT python scripts/eager_host_ab.py r06 2>/dev/null | tail -1 >> $out
done
cat $out
