"""Byte / capacity model of a uniformly partitioned overlap-add reverb against the shipped four-step FFT convolution, at
BASELINE config 4: x (128, 2, 262144), impulse response 65536 taps per (item, channel). Pure arithmetic + the measured per-kernel numbers of
profiles/r04 (reverb_kernel_stats.csv, hbm_traffic_secondary.json); writes profiles/r05/reverb_partitioned_model.json and prints the table
that profiles/r05/reverb_partitioned_model.md quotes.  usage: python scripts/reverb_partition_model.py"""
import csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, C, N, L = 128, 2, 262144, 65536
rows = B * C
cs = rows * N                                   # channel-samples
# ---- what ships (round 4 measurements, same shape, noise generated on the device)
stats = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r04", "reverb_kernel_stats.csv")))}
us = lambda key: sum(v for k, v in stats.items() if key in k)
traffic = json.load(open(os.path.join(ROOT, "profiles", "r04", "hbm_traffic_secondary.json")))["kernels"]
gb = lambda key: sum(v["hbm_bytes"] for k, v in traffic.items() if k.startswith(key)) / 1e9
shipped = {
    "forward_conv_us": us("conv_load_kernel<0") + us("conv_rows_kernel<0") + us("conv_cols_kernel<0"),
    "backward_conv_us": us("conv_load_kernel<1") + us("conv_rows_kernel<1") + us("conv_cols_kernel<1"),
    "ir_spectra_us": us("conv_load_kernel<2") + us("conv_rows_kernel<2") + us("conv_cols_kernel<2"),
    "filter_bank_us": us("fb_fused_kernel") + us("fb_wspectrum_kernel"),
    "forward_conv_GB": gb("conv_load_kernel<0") + gb("conv_rows_kernel<0") + gb("conv_cols_kernel<0"),
    "backward_conv_GB": gb("conv_load_kernel<1") + gb("conv_rows_kernel<1") + gb("conv_cols_kernel<1"),
}
shipped["forward_conv_B_per_sample"] = shipped["forward_conv_GB"] * 1e9 / cs
shipped["forward_conv_TBps"] = shipped["forward_conv_GB"] / shipped["forward_conv_us"] * 1e3
# ---- the chip
CU = {"lds_KiB": 160, "vgpr_KiB": 512, "cus": 256, "l2_MiB_per_xcd": 4, "xcds": 8, "mall_MiB": 256}
on_chip_per_cu = CU["lds_KiB"] + CU["vgpr_KiB"]                      # KiB a workgroup that owns a CU could hold at the very most
# ---- uniformly partitioned overlap-add with block Bk (FFT size 2 Bk, real input: Bk + 1 complex bins ~ 8 Bk bytes per block spectrum)
out = {"shape": [B, C, N], "taps": L, "shipped": shipped, "chip": CU, "partitioned": []}
for Bk in (4096, 8192, 16384, 32768, 65536):
    P = L // Bk                                                      # partitions of the impulse response
    spec = 8 * Bk                                                    # bytes of one block spectrum (complex64, Bk bins)
    H = P * spec                                                     # the response's partition spectra per (item, channel): = 8 L whatever Bk
    hist = (P - 1) * spec                                            # the frequency-domain delay line: the last P - 1 input spectra
    fft_ws = 2 * spec                                                # a 2 Bk-point real transform worked on in place
    need_KiB = (H + hist + fft_ws) / 1024
    blocks = N // Bk
    # (a) everything on the chip: needs H + history + workspace per row on ONE CU
    fits = need_KiB <= on_chip_per_cu
    # (b) "accumulate in registers, re-read X and H from cache" (the verdict's sketch): logical re-reads per output block = P input spectra + P response spectra
    reread_GB = rows * blocks * 2 * P * spec / 1e9
    # (c) three streaming passes (forward block FFTs -> bin-wise delay line over blocks -> inverse FFTs + overlap-add + mix): every spectrum element
    #     is written once and read once in each of its two hops
    b_per_sample = (4 + 8) + (8 + 8) + (8 + 4 + 4)                    # x -> X | X -> Y | Y, x(dry) -> y
    # (d) the delay line on the chip, the response's partition spectra streamed from the last-level cache once per block: footprint = history +
    #     the block being transformed + the accumulator; cache traffic = H per block
    foot_d = (hist + 2 * spec) / 1024
    stream_d = rows * blocks * H / 1e9
    out["partitioned"].append({"variant_d_footprint_KiB": foot_d, "variant_d_fits_one_CU": foot_d <= on_chip_per_cu - 64, "variant_d_cache_stream_GB_forward": stream_d,
                               "block": Bk, "partitions": P, "H_KiB_per_row": H / 1024, "delay_line_KiB_per_row": hist / 1024, "need_KiB_on_one_CU": need_KiB,
                               "fits_one_CU": fits, "variant_b_cache_rereads_GB_forward": reread_GB, "variant_c_B_per_sample_forward": b_per_sample,
                               "variant_c_GB_forward": b_per_sample * cs / 1e9,
                               "variant_c_ms_forward_at_shipped_rate": b_per_sample * cs / 1e9 / shipped["forward_conv_TBps"]})
os.makedirs(os.path.join(ROOT, "profiles", "r05"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r05", "reverb_partitioned_model.json"), "w"), indent=1)
print(f"shipped forward long convolution: {shipped['forward_conv_us']:.0f} us, {shipped['forward_conv_GB']:.2f} GB = {shipped['forward_conv_B_per_sample']:.1f} B per channel-sample "
      f"at {shipped['forward_conv_TBps']:.2f} TB/s;  backward {shipped['backward_conv_us']:.0f} us, {shipped['backward_conv_GB']:.2f} GB;  response spectra {shipped['ir_spectra_us']:.0f} us;  "
      f"filter bank {shipped['filter_bank_us']:.0f} us")
print(f"one CU holds at most {on_chip_per_cu} KiB (LDS {CU['lds_KiB']} + registers {CU['vgpr_KiB']})")
print("block  P   H/row  delay line/row  on one CU?   (b) cache re-reads fwd   (c) 3 streaming passes fwd   (d) delay line on chip, H streamed: footprint, cache traffic fwd")
for r in out["partitioned"]:
    print(f"{r['block']:6d} {r['partitions']:3d} {r['H_KiB_per_row']:6.0f}K {r['delay_line_KiB_per_row']:10.0f}K   {'yes' if r['fits_one_CU'] else 'no ':3s} ({r['need_KiB_on_one_CU']:.0f}K)"
          f"   {r['variant_b_cache_rereads_GB_forward']:10.1f} GB        {r['variant_c_B_per_sample_forward']} B/sample = {r['variant_c_GB_forward']:.2f} GB -> {r['variant_c_ms_forward_at_shipped_rate']:.3f} ms"
          f"   {r['variant_d_footprint_KiB']:.0f}K ({'fits' if r['variant_d_fits_one_CU'] else 'no'}), {r['variant_d_cache_stream_GB_forward']:.1f} GB")
