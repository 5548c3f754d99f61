"""BASELINE config 5 on synthetic data: the reference's style-transfer use case (examples/style_transfer.py in the reference: a network
looks at an input clip and a reference clip and predicts the controls of an EQ -> compressor -> reverb -> gain chain, trained through the
differentiable effects) as a data-parallel training step on MI355X.

One process per GPU (python -m torch.distributed.run --nproc-per-node N examples/style_transfer_synth.py ...): every rank owns its own
batch shard, the effect chain runs on the hand-written HIP kernels of dasp_pytorch_amd with no data-path collective, and the only exchange
is the all-reduce of the networks' gradients (dasp_pytorch_amd.distributed.GradientBuckets: flat buckets that autograd accumulates into,
each all-reduced over RCCL the moment its last gradient arrives, i.e. under the rest of the backward pass).

--model reference (default): the reference's networks at their size - a TCN encoder of ten strided two-convolution blocks (256 channels,
   kernel 7, dilations 1, 2, 4, 8, 16 twice; examples/style_transfer.py:25-87), time-averaged, a 3-layer MLP to a 512-d embedding, and one
   3-layer projector per effect on the concatenated (input, reference) embeddings (:90-127): 10,327,346 parameters = the 41.3 MB
   gradient exchange SURVEY 8(e) sized. The step follows the reference's `step()` (:271-328): the style reference is the input run
   through EQ -> compressor -> reverb with random controls (no gradient: the fused forward kernels), peak-normalised, random gains, both
   clips cut into an A and a B half; the model sees (input A, mono mix of reference B) and is trained on MR-STFT(output A, reference A).
--model small: a ~0.5 M-parameter strided-conv predictor (quick runs and the GPU test suite).

The networks are ordinary PyTorch modules (the user's code around the hot path); the loss is dasp_pytorch_amd.losses' multi-resolution STFT
loss, the fused-kernel counterpart of the auraloss loss the reference trains with.
Prints one JSON line (rank 0): steps/s, clips/s, channel-samples/s through the chain, gradient bytes and buckets, the loss trajectory.
"""
import argparse
import json
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_pytorch_amd as D                      # noqa: E402
from dasp_pytorch_amd import distributed as dd    # noqa: E402

EFFECT_SIZES = (18, 6, 25, 1)                     # EQ, compressor, reverb, gain (dasp_pytorch/modules.py:104-106, 136-155, 179-186, 204-230)


class ControlPredictor(nn.Module):
    """--model small: strided 1-D conv encoder over (input, reference) -> one vector of normalised controls in (0, 1) per effect."""

    def __init__(self, num_controls, width=32):
        super().__init__()
        chans = [2, width, width, 2 * width, 2 * width, 4 * width]
        self.encoder = nn.Sequential(*[
            layer for i in range(5)
            for layer in (nn.Conv1d(chans[i], chans[i + 1], 31, stride=8, padding=15), nn.PReLU(chans[i + 1]))])
        self.head = nn.Sequential(nn.Linear(8 * width, 4 * width), nn.PReLU(), nn.Linear(4 * width, num_controls))

    def forward(self, inp_mono, ref_mono):
        h = self.encoder(torch.cat([inp_mono, ref_mono], 1))
        h = torch.cat([h.mean(-1), h.amax(-1)], 1)
        return torch.sigmoid(self.head(h))


def _mlp(n_in, n_hidden, n_out):
    return [nn.Linear(n_in, n_hidden), nn.ReLU(), nn.Linear(n_hidden, n_hidden), nn.ReLU(), nn.Linear(n_hidden, n_out)]


class ReferenceSizedPredictor(nn.Module):
    """--model reference: the architecture (and so the parameter count and the gradient-exchange volume) of the reference's
    Encoder + four ParameterProjectors, examples/style_transfer.py:25-127. One clip encoder shared by the input and the reference clip;
    its two 512-d embeddings are concatenated and every effect gets its own sigmoid-headed projector."""
    DILATIONS = (1, 2, 4, 8, 16, 1, 2, 4, 8, 16)

    def __init__(self, effect_sizes=EFFECT_SIZES, channels=256, kernel=7, embed=512, hidden=256):
        super().__init__()
        stages, c_in = [], 1
        for d in self.DILATIONS:       # stride-2 dilated convolution, then a plain one; PReLU and batch norm after each (:25-55)
            stages += [nn.Conv1d(c_in, channels, kernel, stride=2, dilation=d), nn.PReLU(channels), nn.BatchNorm1d(channels),
                       nn.Conv1d(channels, channels, kernel), nn.PReLU(channels), nn.BatchNorm1d(channels)]
            c_in = channels
        self.tcn = nn.Sequential(*stages)
        self.to_embedding = nn.Sequential(*_mlp(channels, hidden, embed))
        self.projectors = nn.ModuleList(nn.Sequential(*_mlp(2 * embed, hidden, n), nn.Sigmoid()) for n in effect_sizes)

    def embed(self, clip_mono):
        return self.to_embedding(self.tcn(clip_mono).mean(-1))

    def forward(self, inp_mono, ref_mono):
        z = torch.cat([self.embed(inp_mono), self.embed(ref_mono)], -1)
        return torch.cat([proj(z) for proj in self.projectors], -1)


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


def synth_clips(batch, n, gen, device, channels=2):
    """Speech-like stand-in for the reference's vocal clips: pitched pulse train with a syllable envelope plus noise, peak 0.5."""
    t = torch.arange(n, device=device) / 44100.0
    f0 = 90 + 160 * torch.rand(batch, 1, device=device, generator=gen)
    vib = 1 + 0.02 * torch.sin(2 * torch.pi * 5.5 * t)[None]
    phase = 2 * torch.pi * torch.cumsum(f0 * vib / 44100.0, -1)
    voiced = sum(torch.sin(k * phase) / k for k in range(1, 12))
    env = (0.55 + 0.45 * torch.sin(2 * torch.pi * (2 + 3 * torch.rand(batch, 1, device=device, generator=gen)) * t[None])).clamp_min(0) ** 2
    x = env * (voiced + 0.05 * torch.randn(batch, n, device=device, generator=gen))
    x = 0.5 * x / x.abs().amax(-1, keepdim=True)
    return x[:, None, :].repeat(1, channels, 1).contiguous()


def run(steps=10, batch=8, n=131072, sample_rate=44100, ir_samples=65536, lr=1e-3, width=32, seed=0, quiet=False, graph=False,
        model_kind="small", bucket_mb=16, backend="nccl", force_collectives=False, amp=False):
    """`n` = samples of the section the model is trained on (the reference: 131072 = half of its 262144-sample clips)."""
    rank, local, world = dd.env_world()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dd.init(backend, dev, force=force_collectives)
    torch.manual_seed(seed)                                   # identical network weights on every rank
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)  # different data per rank
    noise_step = torch.zeros(1, dtype=torch.int64, device=dev) if graph else None
    chain = EffectChain(sample_rate, ir_samples, noise_step)
    reference = model_kind == "reference"
    model = (ReferenceSizedPredictor(chain.sizes) if reference else ControlPredictor(chain.num_controls, width)).to(dev)
    whole = graph and world == 1 and not force_collectives    # one GPU: the optimizer step is captured too (no collective in between)
    opt = torch.optim.Adam(model.parameters(), lr=lr if not reference else 1e-4, capturable=whole)      # (the reference trains at 1e-4, :338)
    grads = dd.GradientBuckets(model.parameters(), bucket_bytes=int(bucket_mb * (1 << 20)), force=force_collectives)
    loss_fn = D.losses.MultiResolutionSTFTLoss()              # auraloss' default resolutions, fused HIP kernels
    losses, t0 = [], None
    overlapped = []

    def make_pair(x):
        """-> (model input, training target, clip the networks see as the style reference); no gradients anywhere in here."""
        with torch.no_grad():
            if not reference:                                  # the "style": the same chain with hidden random controls
                target = chain(x, torch.rand(batch, chain.num_controls, device=dev))
                return x, target, target.mean(1, keepdim=True)
            # the reference's step() (examples/style_transfer.py:271-328) on a mono clip of 2 n samples
            ctl = torch.rand(batch, chain.num_controls, device=dev)
            ctl[:, -1] = 0.5                                   # (its gain stage is commented out there, :299: 0 dB)
            ref = chain(x, ctl)
            ref = ref / ref.abs().amax(-1, keepdim=True).clamp_min(1e-8)
            ref = ref * torch.pow(10.0, -torch.rand(batch, 1, 1, device=dev) * 24 / 20)
            inp = x * torch.pow(10.0, -torch.rand(batch, 1, 1, device=dev) * 24 / 20)
            inp_a, ref_a, ref_b = inp[..., :n], ref[..., :n], ref[..., n:]
            return inp_a.contiguous(), ref_a.contiguous(), ref_b.mean(1, keepdim=True)

    def fwd_bwd(x, target, style):
        # amp: the NETWORKS under bf16 autocast (MFMA convolutions; the reference trains them in fp32, and so does the default here: at
        # 8 clips per GPU the fp32 encoder is 98 % of the step). The controls come back as fp32 and the effect chain and the loss always
        # compute in fp32.
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            controls = model(x.mean(1, keepdim=True), style)
        controls = controls.float()
        loss = loss_fn(chain(x, controls), target)
        loss.backward()
        return loss

    def new_clips():
        return synth_clips(batch, 2 * n, gen, dev, 1) if reference else synth_clips(batch, n, gen, dev, 2)

    # --graph: network forward, effect chain, loss and the whole backward pass are captured once into a HIP graph and replayed
    # per step (one launch instead of a few hundred; at 8 clips per GPU the step is launch-bound otherwise). The hand-written
    # kernels are plain stream launches on torch's current stream, so they are captured like any torch op. On one GPU the target
    # chain and the optimizer step are part of the graph as well; with several, the gradient all-reduce and the optimizer stay eager
    # (the hooks of GradientBuckets are silent during a capture; finish() issues the collectives after the replay).
    g, static = None, {}
    if graph:
        static["clips"] = new_clips()
        static["pair"] = [torch.zeros_like(t) for t in make_pair(static["clips"])]

        def captured():
            grads.zero_grad()
            pair = make_pair(static["clips"]) if whole else static["pair"]
            loss = fwd_bwd(*pair)
            if whole:
                opt.step()
            noise_step.add_(1)                                # captured too: the next replay's reverbs draw different noise
            return loss
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), grads.no_sync():        # warm-up off the capture stream (allocator, lazy tables, Adam state); no
            for _ in range(3):                                # collective from these passes: their gradients are thrown away
                captured()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static["loss"] = captured()

    for step in range(steps + 1):                             # step 0 warms the caches / clocks and is not timed
        if step == 1:
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            t0 = time.perf_counter()
        clips = new_clips()
        if g is None:
            pair = make_pair(clips)
            grads.zero_grad()
            loss = fwd_bwd(*pair)
        else:
            static["clips"].copy_(clips)
            if not whole:
                for dst, src in zip(static["pair"], make_pair(clips)):
                    dst.copy_(src)
            g.replay()                                        # gradients land in the flat buckets the capture accumulated into
            loss = static["loss"]
        if not whole:
            grads.finish()                                    # the one collective of the job: waits for the buckets launched under backward
            overlapped.append(grads.launched_in_backward)
            opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = dd.max_over_ranks(time.perf_counter() - t0, dev) / max(steps, 1)
    out = {"workload": "style-transfer chain EQ->compressor->reverb->gain, networks + MR-STFT loss, data parallel", "n_gpus": world,
           "model": model_kind, "networks_dtype": "bf16 autocast" if amp else "f32", "parameters": sum(p.numel() for p in model.parameters()),
           "gradient_bytes_per_step": grads.bytes, "gradient_buckets": len(grads.buckets), "collectives_active": bool(grads.active),
           "buckets_launched_under_backward": overlapped[-1] if overlapped else 0,
           "clip": [batch, 1 if reference else 2, n], "ir_samples": ir_samples, "steps": steps, "s_per_step": dt, "clips_per_s": world * batch / dt,
           "channel_samples_per_s": world * batch * 2 * n / dt, "hip_graph": bool(graph), "loss_first": losses[0], "loss_last": losses[-1],
           "finite": bool(all(map(lambda v: v == v and abs(v) != float("inf"), losses)))}
    if rank == 0 and not quiet:
        print(json.dumps(out))
    grads.remove()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return out, model


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--samples", type=int, default=131072, help="samples of the section the model is trained on")
    ap.add_argument("--ir-samples", type=int, default=65536)
    ap.add_argument("--model", choices=("reference", "small"), default="reference")
    ap.add_argument("--bucket-mb", type=float, default=16.0, help="gradient bucket size (MiB)")
    ap.add_argument("--graph", action="store_true", help="capture forward + backward of the step into a HIP graph and replay it")
    ap.add_argument("--force-collectives", action="store_true", help="run the gradient all-reduce through RCCL on a one-rank group too")
    ap.add_argument("--amp", action="store_true", help="run the networks (not the effects, not the loss) under bf16 autocast")
    a = ap.parse_args()
    run(a.steps, a.batch, a.samples, ir_samples=a.ir_samples, graph=a.graph, model_kind=a.model, bucket_mb=a.bucket_mb,
        force_collectives=a.force_collectives, amp=a.amp)
