"""Developer probe: which part of a captured parametric_eq step upsets hipStreamEndCapture (each variant in its own process)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch, numpy as np
    import dasp_pytorch_amd as D
    from dasp_pytorch_amd import config
    mode = sys.argv[1]
    config.plan.torch_ops = len(sys.argv) > 2 and sys.argv[2] == "1"           # the binding under test (the probe was written for the ctypes one)
    B, C, N = 8, 2, 131072
    g = torch.Generator(device="cuda:0").manual_seed(5)
    xs = (torch.rand(B, C, N, device="cuda:0", generator=g) * 2 - 1).requires_grad_(True)
    rng = np.random.default_rng(1)
    lo = [-20, 20, .1] * 6; hi = [20, 20000, 6] * 6
    lo[1], hi[1] = 20, 2000; lo[16], hi[16] = 4000, 20000
    cols = [torch.tensor(rng.uniform(lo[i], hi[i], B).astype("float32"), device="cuda:0").requires_grad_(True) for i in range(18)]
    w = torch.randn(B, C, N, device="cuda:0", generator=g)
    if "pre" in mode:
        D.parametric_eq(xs, 44100, *cols).backward(w)
    if "retain" in mode:
        y0 = D.parametric_eq(xs, 44100, *cols)
        y0.backward(w, retain_graph=True)
        y0.backward(w, retain_graph=True)
    if "del" in mode:
        del y0
    if "puretorch" in mode:
        f = lambda: (xs * cols[0].view(-1, 1, 1)).cumsum(-1) * cols[1].view(-1, 1, 1)
        y1 = f()
        y1.backward(w, retain_graph=True)
        y1.backward(w, retain_graph=True)
        D.parametric_eq = lambda xs, sr, *c: f()
    if "once" in mode:
        y2 = D.parametric_eq(xs, 44100, *cols)
        y2.backward(w, retain_graph=True)
    if "nobwd" in mode:
        y3 = D.parametric_eq(xs, 44100, *cols)
    if "gen" in mode:
        xn = torch.rand(B, C, N, device="cuda:0", generator=g)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            D.parametric_eq(xs, 44100, *cols).backward(w)
    torch.cuda.current_stream().wait_stream(s)
    if "none" in mode:
        xs.grad = None
        for c in cols: c.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ys = D.parametric_eq(xs, 44100, *cols)
        if "fwd" not in mode:
            if "autograd" in mode:
                gs = torch.autograd.grad(ys, [xs] + cols, w)
            else:
                ys.backward(w)
    graph.replay(); torch.cuda.synchronize()
    print(mode, "ok", float(ys.abs().sum()))
else:
    for env, mode in [("0", "bwd-none-retain-del"), ("0", "bwd-none-puretorch"), ("0", "bwd-none-once"), ("0", "bwd-none-nobwd")]:
        r = subprocess.run([sys.executable, __file__, mode, env], capture_output=True, text=True, timeout=300)
        print("config.plan.torch_ops=" + env, mode, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], "|", [l for l in r.stderr.splitlines() if "Error" in l or "error" in l][:3])
