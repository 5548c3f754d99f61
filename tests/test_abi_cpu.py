"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU, exports exactly
the symbols include/dasp_hip.h declares, answers its pure size queries, and the Python product
layer refuses CPU tensors loudly (there is no CPU fallback)."""
import os
import re

import pytest
import torch

from dasp_pytorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    h = open(os.path.join(ROOT, "include", "dasp_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(dasp_\w+)\s*\(", h)))


def test_header_and_binding_agree():
    assert _header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    for name in _header_symbols():
        assert hasattr(L, name), name


def test_size_queries():
    L = _lib.lib()
    assert L.dasp_sos_tile() == 64 * L.dasp_sos_chunk()
    for S in (2, 4, 6, 8):
        assert L.dasp_sos_supported_sections(S) == 1
        assert L.dasp_sos_table_floats(S) > 0
    assert L.dasp_sos_supported_sections(5) == 0
    assert L.dasp_sos_table_floats(5) == -1
    T = L.dasp_sos_tile()
    assert L.dasp_sos_num_tiles(1) == 1 and L.dasp_sos_num_tiles(T) == 1 and L.dasp_sos_num_tiles(T + 1) == 2
    assert L.dasp_sos_carry_floats(4, 2 * T, 6) == 4 * 2 * 12 * 64


def test_segment_planning():
    """Host-side planning of the segmented-row path (no launch): few rows of a long signal are cut into power-of-two runs of at least
    eight tiles so that rows * segments fills the chip; many rows or short signals keep one workgroup per row."""
    L = _lib.lib()
    T = L.dasp_sos_tile()
    N = 128 * T
    assert L.dasp_sos_segment_tiles(512, N) == 0 and L.dasp_sos_segment_tiles(129, N) == 0      # enough rows (129 .. 256: twice the waves per row instead)
    assert L.dasp_sos_segment_tiles(8, 15 * T) == 0                                             # too short to cut
    for rows in (1, 2, 16, 32, 64, 65, 96, 128):
        t = L.dasp_sos_segment_tiles(rows, N)
        g = L.dasp_sos_segments(N, t)
        assert t >= 8 and t & (t - 1) == 0 and g == -(-128 // t) and g > 1 and rows * g <= 1024
        assert L.dasp_sos_seg_floats(rows, N, 6, t) == 2 * rows * g * 12
    assert L.dasp_sos_segments(N + 1, 8) == 17 and L.dasp_sos_segments(N, 0) == 1                # ragged last segment; 0 = not segmented
    # per item: the two segment transition matrices + the basis responses of the Gram finalize step (28 columns: 7 blocks of 16 signal rows
    # for the matrix-core products + 2 x 6 x 16 adjoint rows)
    assert L.dasp_sos_segtab_doubles(6) == 2 * 12 * 12 + 28 * (7 * 16 + 2 * 6 * 16)
    assert L.dasp_sos_partial_floats(4 * L.dasp_sos_segments(N, 8), 6) == 16 * L.dasp_sos_partial_floats(4, 6)


def test_reverb_size_query():
    """Host-side planning of the reverb (no launch): block length, transform length, pairs; refusals are -2."""
    import ctypes
    L = _lib.lib()
    sizes = (ctypes.c_long * 14)()
    assert L.dasp_reverb_sizes(128, 262144, 65536, 1023, 12, sizes) == 0
    Lb, n1, pairs, nblk = sizes[0], sizes[1], sizes[2], sizes[3]
    assert (Lb, n1, pairs, nblk) == (65536, 131072, 2, 4)
    assert sizes[4] == 13 * 4096 + 12 * 1023 // 2 and sizes[5] == -(-65536 // 3072)   # twiddles + 12 band spectra + the taps; 3072 valid samples per window
    assert sizes[6] == 256 * pairs * n1 and sizes[7] == 128 * n1 and sizes[8] == 256 * 65536
    chunk = sizes[9]                                                          # signals per pass of the long-convolution pipeline
    assert chunk == 256 and sizes[12] == chunk * pairs * n1 and sizes[13] == chunk // 2 * n1   # one pass over all signals by default
    assert L.dasp_reverb_sizes(1, 9000, 1000, 63, 12, sizes) == 0 and (sizes[0], sizes[3], sizes[2]) == (4096, 3, 2)   # minimum block; odd block count: zero partner
    assert sizes[12] == 1 * 12 * 4096                                         # small problems: W / Ag sized by the per-item band spectra they also hold
    assert L.dasp_reverb_sizes(1, 1000, 4096, 3587, 12, sizes) == -2         # filter longer than the filter-bank window
    assert L.dasp_reverb_sizes(1, 1000, (1 << 20) + 1, 63, 12, sizes) == -2  # impulse response beyond 2^20 samples
    assert L.dasp_reverb_sizes(1, 1000, 4096, 63, 17, sizes) == -1           # more than 16 bands


def test_argument_errors_without_gpu():
    """NULL pointers / bad sizes are rejected before any launch (status DASP_ERR_ARG = -1)."""
    L = _lib.lib()
    assert L.dasp_sosfilt_forward(None, 1, None, None, None, 1, 1, 16, 6, None) == -1
    assert L.dasp_sos_prepare(None, 1, 6, None, None, None) == -1
    assert L.dasp_lfilter_forward(None, None, None, 1, None, None, None, 0, 1, 16, 5, 0, 0, None) == -1
    assert L.dasp_lfilter_work_doubles(4, 262144, 5, 0) == 2 * 512 * 4 * 7 + 4 * 7 * 7          # up to 64 rows: 512 chunks of 512 samples, 7 state components (K <= 8)
    assert L.dasp_lfilter_work_doubles(100, 262144, 5, 0) == 2 * 256 * 100 * 7 + 100 * 7 * 7    # more rows: 256 chunks of 1024 samples
    assert L.dasp_lfilter_work_doubles(4, 50, 5, 0) == 0 and L.dasp_lfilter_work_doubles(4, 100, 17, 0) == -1   # one chunk (<= 64 samples): no scratch; K > 16 unsupported
    assert L.dasp_lfilter_work_doubles(4, 100, 5, 7) == 2 * 15 * 4 * 7 + 4 * 7 * 7               # the caller's chunk length: 15 chunks of 7 samples
    # a scratch buffer smaller than the plan needs is refused before anything is launched (the pointers are never dereferenced on the host)
    assert L.dasp_lfilter_forward(8, 8, 8, 1, 8, None, 8, 10, 4, 262144, 5, 0, 0, None) == -1
    assert L.dasp_mrstft_backward_target(None, None, None, None, None, None, 1, 4096, 0, None, None, None, 1e-8, None) == -1


def test_product_has_no_cpu_path():
    import dasp_pytorch_amd as D
    x = torch.zeros(1, 1, 64)
    p = [torch.ones(1)] * 18
    with pytest.raises(_lib.DaspHipError):
        D.parametric_eq(x, 44100, *p)
    with pytest.raises(_lib.DaspHipError):
        D.signal.sosfilt_via_fsm(torch.zeros(1, 2, 6), x)


def test_tensors_of_one_call_must_share_a_device():
    """_lib.require_same_device (every op calls it before handing raw pointers to the library): a tensor on another device than x is
    refused with both devices named. Host logic - torch's meta device stands in for a second GPU."""
    x = torch.zeros(2, 2, 8)
    _lib.require_same_device(x, sos=torch.zeros(2, 1, 6), nothing=None, number=3.0)           # same device, non-tensors: fine
    with pytest.raises(_lib.DaspHipError, match="sos is on meta but x is on cpu"):
        _lib.require_same_device(x, sos=torch.zeros(2, 1, 6, device="meta"))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "dasp_pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_dynamics_segment_planning():
    """Host-side planning of the segmented compressor path (no launch): few items of a long signal are cut into power-of-two runs of at
    least one tile per forward wave so that items * segments fills the chip; many items or short signals keep one workgroup per item."""
    L = _lib.lib()
    T = 512                                                    # samples per compressor tile
    assert L.dasp_dyn_num_tiles(T) == 1 and L.dasp_dyn_num_tiles(T + 1) == 2
    N = 512 * T
    assert L.dasp_dyn_segment_tiles(256, N) == 0 and L.dasp_dyn_segment_tiles(128, N) == 0      # enough items
    assert L.dasp_dyn_segment_tiles(8, 31 * T) == 0                                              # too short to cut
    for B in (1, 2, 8, 16, 32, 100):
        t = L.dasp_dyn_segment_tiles(B, N)
        g = L.dasp_dyn_segments(N, t)
        assert t >= 16 and t & (t - 1) == 0 and g == -(-512 // t) and g > 1 and B * g <= 1024
    assert L.dasp_dyn_segments(N + 1, 16) == 33 and L.dasp_dyn_segments(N, 0) == 1
    assert L.dasp_dyn_partial_floats(4 * 33) == 33 * L.dasp_dyn_partial_floats(4)
