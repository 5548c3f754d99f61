#!/bin/bash
out=gpurun_out/r05r; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_reverb.py tests/test_gpu_chain.py tests/test_gpu_graph_replay.py -q -m gpu --tb=short 2>&1 | tail -2
timeout 400 python scripts/fb_split_sweep.py 2>/dev/null | grep -E "split=plan|split=1\"" | tee $out/fb_split_rule.log
timeout 300 python scripts/chain_graph_time.py 2>/dev/null | tail -1 | tee -a $out/fb_split_rule.log
