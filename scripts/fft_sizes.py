"""Times batched R2C + C2R (torch.fft = hipFFT/rocFFT, the same kernels reverb.hip's plans run) for candidate
transform lengths, to choose the padded lengths of the reverb's two convolutions. Developer tool."""
import sys, torch
dev = "cuda"
def t(rows, n, it=10):
    x = torch.randn(rows, n, device=dev)
    for _ in range(3):
        X = torch.fft.rfft(x); y = torch.fft.irfft(X, n=n)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        X = torch.fft.rfft(x); y = torch.fft.irfft(X, n=n)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
x = torch.randn(4096, 65536, device=dev)
for _ in range(200): x = x * 1.0001      # clock ramp
for rows, ns in ((3072, (131072, 73728, 69632, 66560, 67584, 81920, 98304)), (1024, (131072,)), (256, (524288, 327680, 331776, 393216)), (512, (262144, 163840))):
    for n in ns:
        ms = t(rows, n)
        print(f"rows {rows:5d} n {n:7d}: r2c+c2r {ms:7.3f} ms  ({rows*n*4*4/ms/1e6:7.1f} GB/s on 16 B/sample)", flush=True)
