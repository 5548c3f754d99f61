"""BASELINE config 5 on synthetic data: the reference's style-transfer use case (examples/style_transfer.py in the reference:
a network looks at an input clip and a reference clip and predicts the controls of an EQ -> compressor -> reverb chain, trained
through the differentiable effects) as a data-parallel training step on MI355X.

One process per GPU (python -m torch.distributed.run --nproc-per-node N examples/style_transfer_synth.py ...): every rank owns its
own batch shard, the effect chain runs on the hand-written HIP kernels of dasp_pytorch_amd with no data-path collective, and the only
exchange is the bucketed all-reduce of the predictor's gradients (dasp_pytorch_amd.distributed.allreduce_gradients, RCCL over xGMI).
The predictor is an ordinary PyTorch module (the user's code around the hot path); the loss is dasp_pytorch_amd.losses'
multi-resolution STFT loss, the fused-kernel counterpart of the auraloss loss the reference trains with.

Prints one JSON line (rank 0): steps/s, clips/s, channel-samples/s through the chain, the loss trajectory.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D                      # noqa: E402
from dasp_pytorch_amd import distributed as dd    # noqa: E402


class ControlPredictor(torch.nn.Module):
    """Strided 1-D conv encoder over (input, reference) -> one vector of normalised controls in (0, 1) per effect."""

    def __init__(self, num_controls, width=32):
        super().__init__()
        chans = [2, width, width, 2 * width, 2 * width, 4 * width]
        self.encoder = torch.nn.Sequential(*[
            layer for i in range(5)
            for layer in (torch.nn.Conv1d(chans[i], chans[i + 1], 31, stride=8, padding=15), torch.nn.PReLU(chans[i + 1]))])
        self.head = torch.nn.Sequential(torch.nn.Linear(8 * width, 4 * width), torch.nn.PReLU(), torch.nn.Linear(4 * width, num_controls))

    def forward(self, inp_mono, ref_mono):
        h = self.encoder(torch.cat([inp_mono, ref_mono], 1))
        h = torch.cat([h.mean(-1), h.amax(-1)], 1)
        return torch.sigmoid(self.head(h))


class EffectChain:
    """EQ -> compressor -> reverb -> gain, controls normalised to (0, 1): the reference's StyleTransferModel wiring
    (examples/style_transfer.py:150-154) on dasp_pytorch_amd.chain.StyleTransferChain, which folds the gain into the compressor."""

    def __init__(self, sample_rate, ir_samples=65536, noise_seed_offset=None):
        # device_noise: the reverb's white noise is generated inside its filter-bank kernels. Its seed is a host draw per call, frozen when a
        # launch is captured into a HIP graph: the offset word (a device int64 the step bumps) gives every replay new noise
        self.chain = D.chain.StyleTransferChain(sample_rate, num_samples=ir_samples, device_noise=True, noise_seed_offset=noise_seed_offset)
        self.sizes = self.chain.num_params

    @property
    def num_controls(self):
        return sum(self.sizes)

    def __call__(self, x, controls):
        return self.chain.process_normalized(x, *torch.split(controls, self.sizes, dim=1))


def synth_clips(batch, n, gen, device):
    """Speech-like stand-in for the reference's vocal clips: pitched pulse train with a syllable envelope plus noise, peak 0.5."""
    t = torch.arange(n, device=device) / 44100.0
    f0 = 90 + 160 * torch.rand(batch, 1, device=device, generator=gen)
    vib = 1 + 0.02 * torch.sin(2 * torch.pi * 5.5 * t)[None]
    phase = 2 * torch.pi * torch.cumsum(f0 * vib / 44100.0, -1)
    voiced = sum(torch.sin(k * phase) / k for k in range(1, 12))
    env = (0.55 + 0.45 * torch.sin(2 * torch.pi * (2 + 3 * torch.rand(batch, 1, device=device, generator=gen)) * t[None])).clamp_min(0) ** 2
    x = env * (voiced + 0.05 * torch.randn(batch, n, device=device, generator=gen))
    x = 0.5 * x / x.abs().amax(-1, keepdim=True)
    return x[:, None, :].repeat(1, 2, 1).contiguous()


def run(steps=10, batch=8, n=131072, sample_rate=44100, ir_samples=65536, lr=1e-3, width=32, seed=0, quiet=False, graph=False):
    rank, local, world = dd.env_world()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dd.init("nccl", dev)
    torch.manual_seed(seed)                                   # identical predictor weights on every rank
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)  # different data per rank
    noise_step = torch.zeros(1, dtype=torch.int64, device=dev) if graph else None
    chain = EffectChain(sample_rate, ir_samples, noise_step)
    model = ControlPredictor(chain.num_controls, width).to(dev)
    whole = graph and world == 1                              # one GPU: the optimizer step is captured too (no collective in between)
    opt = torch.optim.Adam(model.parameters(), lr=lr, capturable=whole)
    loss_fn = D.losses.MultiResolutionSTFTLoss()              # auraloss' default resolutions, fused HIP kernels
    losses, t0 = [], None

    def fwd_bwd(x, target):
        controls = model(x.mean(1, keepdim=True), target.mean(1, keepdim=True))
        loss = loss_fn(chain(x, controls), target)
        loss.backward()
        return loss

    def make_target(x):                                       # the "style": the same chain with hidden random controls
        with torch.no_grad():
            return chain(x, torch.rand(batch, chain.num_controls, device=dev))

    # --graph: predictor forward, effect chain, loss and the whole backward pass are captured once into a HIP graph and replayed
    # per step (one launch instead of a few hundred; at 8 clips per GPU the step is launch-bound otherwise). The hand-written
    # kernels are plain stream launches on torch's current stream, so they are captured like any torch op. On one GPU the target
    # chain and the optimizer step are part of the graph as well; with several, the gradient all-reduce and the optimizer stay eager.
    g, static = None, {}
    if graph:
        static["x"] = synth_clips(batch, n, gen, dev)
        static["target"] = torch.zeros_like(static["x"])

        def captured():
            if whole:
                static["target"] = make_target(static["x"])
            loss = fwd_bwd(static["x"], static["target"])
            if whole:
                opt.step()
            noise_step.add_(1)                                # captured too: the next replay's reverbs draw different noise
            return loss
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                         # warm-up off the capture stream (allocator, lazy tables, Adam state)
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                captured()
        torch.cuda.current_stream().wait_stream(side)
        opt.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static["loss"] = captured()

    for step in range(steps + 1):                             # step 0 warms the caches / clocks and is not timed
        if step == 1:
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            t0 = time.perf_counter()
        x = synth_clips(batch, n, gen, dev)
        if g is None:
            target = make_target(x)
            opt.zero_grad(set_to_none=True)
            loss = fwd_bwd(x, target)
        else:
            static["x"].copy_(x)
            if not whole:
                static["target"].copy_(make_target(x))
            g.replay()                                        # gradients land in the .grad tensors the capture allocated
            loss = static["loss"]
        if not whole:
            dd.allreduce_gradients(model.parameters())        # the one collective of the job
            opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = dd.max_over_ranks(time.perf_counter() - t0, dev) / max(steps, 1)
    out = {"workload": "style-transfer chain EQ->compressor->reverb, predictor + MR-STFT loss, data parallel", "n_gpus": world,
           "clip": [batch, 2, n], "ir_samples": ir_samples, "steps": steps, "s_per_step": dt, "clips_per_s": world * batch / dt,
           "channel_samples_per_s": world * batch * 2 * n / dt, "hip_graph": bool(graph), "loss_first": losses[0], "loss_last": losses[-1],
           "finite": bool(all(map(lambda v: v == v and abs(v) != float("inf"), losses)))}
    if rank == 0 and not quiet:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()
    return out, model


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--samples", type=int, default=131072)
    ap.add_argument("--ir-samples", type=int, default=65536)
    ap.add_argument("--graph", action="store_true", help="capture forward + backward of the step into a HIP graph and replay it")
    a = ap.parse_args()
    run(a.steps, a.batch, a.samples, ir_samples=a.ir_samples, graph=a.graph)
