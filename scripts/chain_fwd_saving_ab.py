"""Round-6 A/B for SURVEY 8(f2) on the pass that carries gradients: EQ -> compressor forward as the training step runs it (two launches
that save what their backward passes read: 4 + 4 + 3 and 4 + 4 B per channel-sample = 19 B) against ONE fused pass that saves the same
(dasp_chain_forward_saving, a prototype: x in, y out, the EQ's output for the compressor's backward recompute, the EQ's chunk states =
15 B). Graph replays, blocks interleaved, outputs compared. Measured and closed (profiles/r06/chain_fwd_saving_ab.log): the prototype entry point
exists at commit e745b07 only - check that commit out to run this script.
usage: python scripts/chain_fwd_saving_ab.py [B C N]"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dasp_pytorch_amd import _lib
from dasp_pytorch_amd._lib import ptr, stream
from dasp_pytorch_amd.functional import _PEQ_TYPES

B, C, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 2, 131072)
S, SR, dev = 6, 44100.0, "cuda"
L = _lib.lib()
L.dasp_chain_forward_saving.restype = ctypes.c_int
L.dasp_chain_forward_saving.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_void_p]
R = [(-20, 20), (20, 2000), (0.1, 6), (-20, 20), (80, 2000), (0.1, 6), (-20, 20), (2000, 8000), (0.1, 6),
     (-20, 20), (8000, 12000), (0.1, 6), (-20, 20), (12000, 21050), (0.1, 6), (-20, 20), (4000, 21050), (0.1, 6)]
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, N, device=dev, generator=g) * 2 - 1
cols = [(torch.rand(B, device=dev, generator=g) * (hi - lo) + lo).contiguous() for lo, hi in R]
rows = (ctypes.c_void_p * 18)(*[c.data_ptr() for c in cols])
tys = (ctypes.c_int * 6)(*_PEQ_TYPES)
dyn = [(-60, 0), (1, 20), (5, 100), (1e-3, 12), (0, 12)]
ctl = torch.stack([torch.rand(B, device=dev, generator=g) * (hi - lo) + lo for lo, hi in dyn], 1).contiguous()
tab = torch.empty(B * L.dasp_sos_table_floats(S), device=dev)
dtab = torch.empty(B * L.dasp_sos_dtab_doubles(S), dtype=torch.float64, device=dev)
car = torch.empty(L.dasp_sos_carry_floats(B * C, N, S), device=dev)
car2 = torch.empty_like(car)
dcar = torch.empty(L.dasp_dyn_carry_floats(B, N), device=dev)
yeq, y, yeq2, y2 = (torch.empty_like(x) for _ in range(4))


def separate():
    _lib.call("dasp_peq_forward", rows, B, S, tys, SR, ptr(tab), ptr(dtab), ptr(x), ptr(yeq), ptr(car), B, C, N, 0, ptr(None), ptr(None), stream())
    _lib.call("dasp_dynamics_forward", 0, ptr(yeq), ptr(ctl), ptr(y), ptr(dcar), ptr(None), B, C, N, SR, ctypes.c_float(1e-8), 0, stream())


def fused():
    _lib.call("dasp_peq_prepare_rows", rows, B, S, tys, SR, ptr(tab), ptr(dtab), stream())
    _lib.check(L.dasp_chain_forward_saving(tab.data_ptr(), B, x.data_ptr(), ctl.data_ptr(), y2.data_ptr(), yeq2.data_ptr(), car2.data_ptr(), B, C, N, S, 0, SR, 1e-8,
                                           torch.cuda.current_stream().cuda_stream), "dasp_chain_forward_saving")


separate(); fused(); torch.cuda.synchronize()
err = {"y": float((y - y2).abs().max() / y.abs().max()), "yeq": float((yeq - yeq2).abs().max() / yeq.abs().max()),
       "states": float((car - car2).abs().max() / car.abs().max())}
graphs = {}
side = torch.cuda.Stream()
for name, fn in (("separate", separate), ("fused", fused)):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        fn()
    graphs[name] = gph
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for gph in graphs.values():
        for _ in range(10):
            gph.replay()
    torch.cuda.synchronize()
times = {"separate": [], "fused": []}
for blk in range(8):
    for name in (("separate", "fused") if blk % 2 == 0 else ("fused", "separate")):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            graphs[name].replay()
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / 100 * 1e3)
cs = B * C * N
print(json.dumps({"shape": [B, C, N], "separate_ms": round(float(np.median(times["separate"])), 4), "fused_saving_ms": round(float(np.median(times["fused"])), 4),
                  "separate_bytes": 19 * cs, "fused_bytes": 15 * cs, "separate_TBps": round(19 * cs / np.median(times["separate"]) / 1e9, 2),
                  "fused_TBps": round(15 * cs / np.median(times["fused"]) / 1e9, 2), "rel_diff": err}))
