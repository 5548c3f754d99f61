"""Raw timing of the device random stream's kernels with a given build of libdasp_hip.so (no self-check, no comparison: for probe builds
whose results are wrong on purpose). usage: python scripts/mtrand_raw.py LIB.so [bs ...]; run under rocprofv3 --kernel-trace --stats."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dasp_pytorch_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from dasp_pytorch_amd import _mt19937 as mt

rng = np.random.default_rng(0)
for bs in [int(a) for a in sys.argv[2:]] or [8]:
    out = torch.empty(2 * bs * 12 * 66558, device="cuda:0")
    words = rng.integers(0, 1 << 32, 624, dtype=np.uint64).astype(np.uint32)
    for _ in range(6):
        mt._randn_from_state(words, 17, out)
    torch.cuda.synchronize()
    print(bs, "done", float(out[:1024].float().abs().mean()))
