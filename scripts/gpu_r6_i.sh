#!/bin/bash
# round 6: the fused EQ -> compressor training forward (torch.ops.dasp.eq_dyn_norm) - parity tests, chain step A/B, whole suite
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_torch_ops.py -q 2>&1 | tail -8 | tee $out/pytest_chain.log
timeout 600 python scripts/chain_step_ab.py 256 2 131072 256 1 131072 192 2 131072 128 2 131072 2>&1 | tail -4 | tee $out/chain_step_ab.log

