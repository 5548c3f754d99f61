"""Same-box A/B at the north-star shape (256,2,131072): the EQ step as the product launches it (design, forward, backward, finalize:
four launches) against the step whose backward launch finalizes (dasp_peq_forward / dasp_peq_backward with Tseg = -1: the basis responses
come out of the design launch, every row's workgroup turns its Gram matrix into lag sums and the last one of an item maps them to the 18
control gradients - three launches). Both as captured graphs of the two C calls, blocks interleaved; outputs compared.
The Tseg = -1 path was measured and dropped (profiles/r06/fused_finalize_ab.log: +13 us, bit-identical results): it exists at commit 3fe6843 only -
check that commit out to run this script.   usage: python scripts/fused_finalize_ab.py [B C N]"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dasp_pytorch_amd import _lib
from dasp_pytorch_amd._lib import call, ptr, stream
from dasp_pytorch_amd.functional import _PEQ_TYPES

B, C, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 2, 131072)
S, SR, dev = 6, 44100.0, "cuda"
L = _lib.lib()
R = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
     (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, N, device=dev, generator=g) * 2 - 1
w = torch.randn(B, C, N, device=dev, generator=g)
cols = [(torch.rand(B, device=dev, generator=g) * (hi - lo) + lo).contiguous() for lo, hi in R]
rows = (ctypes.c_void_p * 18)(*[c.data_ptr() for c in cols])
tys = (ctypes.c_int * 6)(*_PEQ_TYPES)
tab = torch.empty(B * L.dasp_sos_table_floats(S), device=dev)
dtab = torch.empty(B * L.dasp_sos_dtab_doubles(S), dtype=torch.float64, device=dev)
segtab = torch.empty(B * L.dasp_sos_segtab_doubles(S), dtype=torch.float64, device=dev)
car = torch.empty(L.dasp_sos_carry_floats(B * C, N, S), device=dev)
part = torch.empty(L.dasp_sos_partial_floats(B * C, S), device=dev)
y, gx, gout = torch.empty_like(x), torch.empty_like(x), torch.empty(18, B, device=dev)


def step(tseg):
    call("dasp_peq_forward", rows, B, S, tys, SR, ptr(tab), ptr(dtab), ptr(x), ptr(y), ptr(car), B, C, N, tseg, ptr(segtab if tseg else None), ptr(None), stream())
    call("dasp_peq_backward", ptr(tab), ptr(dtab), B, ptr(x), ptr(w), ptr(car), ptr(gx), ptr(part), 2, ptr(gout), B, C, N, S, tseg,
         ptr(segtab if tseg else None), ptr(None), stream())


res = {}
for tseg in (0, -1):
    step(tseg)
    torch.cuda.synchronize()
    res[tseg] = (y.clone(), gx.clone(), gout.clone())
ey = float((res[0][0] - res[-1][0]).abs().max())
egx = float((res[0][1] - res[-1][1]).abs().max() / res[0][1].abs().max())
eg = float(((res[0][2] - res[-1][2]).abs().amax(1) / res[0][2].abs().amax(1)).max())
graphs = {}
side = torch.cuda.Stream()
for tseg in (0, -1):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(tseg)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        step(tseg)
    graphs[tseg] = gph
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for gph in graphs.values():
        for _ in range(10):
            gph.replay()
    torch.cuda.synchronize()
times = {0: [], -1: []}
for blk in range(8):
    for tseg in ((0, -1) if blk % 2 == 0 else (-1, 0)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            graphs[tseg].replay()
        torch.cuda.synchronize()
        times[tseg].append((time.perf_counter() - t0) / 100 * 1e3)
print(json.dumps({"shape": [B, C, N], "four_launches_ms": round(float(np.median(times[0])), 4), "backward_finalizes_ms": round(float(np.median(times[-1])), 4),
                  "blocks_four": [round(t, 4) for t in times[0]], "blocks_fused": [round(t, 4) for t in times[-1]],
                  "y_max_abs_diff": ey, "gx_rel_diff": egx, "control_gradients_rel_diff": eg}))
