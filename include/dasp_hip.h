/* dasp_hip.h -- C ABI of libdasp_hip.so: the MI355X (gfx950) hot path of dasp_pytorch.functional.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream (hipStream_t passed as
 * void*; NULL = the null stream). All launches are asynchronous on that stream, out-of-place, and
 * never allocate: the caller owns every buffer (sizes come from the *_floats / *_doubles queries).
 * Return value: 0 = success, >0 = hipError_t of the failed launch, <0 = DASP_ERR_*.
 *
 * The reference (csteinmetz1/dasp-pytorch v0.0.1) has no FFI layer -- its boundary is the Python
 * function API. Each group below names the reference callable (file:line) it replaces; the Python
 * binding that restores the reference signatures is dasp_pytorch_amd/ (see INTEGRATION.md).
 */
#ifndef DASP_HIP_H
#define DASP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* Hash of this header as it was when the library was built (csrc/build.py -> csrc/abi.hip). A binding compiled against the header - the
 * torch extension csrc/torch_ext, or a maintainer's own - compares it with the hash it was built with before it trusts the argument lists. */
unsigned long long dasp_abi_hash(void);

#define DASP_OK 0
#define DASP_ERR_ARG (-1)          /* null pointer / inconsistent sizes */
#define DASP_ERR_UNSUPPORTED (-2)  /* e.g. section count without a compiled kernel */
#define DASP_ERR_DEVICE (-3)       /* a kernel of an EARLIER call on this device reported a broken protocol: see dasp_device_error below */

/* ---------------------------------------------------------------------------------------------
 * Device-side errors and the look-back plan.  The segmented launches ("Few rows", "Few batch items") hand segment states between the
 * workgroups of ONE launch as tagged 64-bit words; a reader that waits for a word longer than 2 s of wall clock (a scratch buffer
 * overwritten under the launch, a workgroup that died) stores into a sticky error word of its kernel family and carries on with NaN.
 * The words live in 64 bytes of host-mapped memory per device (one allocation, made by the first call of dasp_device_error() or of a
 * segmented entry point on that device - call dasp_device_error() once before capturing a graph); nothing is copied or polled on the
 * fast path. While a bit is set every segmented entry point returns DASP_ERR_DEVICE instead of launching.
 *   dasp_device_error()        bits of the current device: 1 EQ forward, 2 EQ backward, 4 compressor/expander forward, 8 backward, 16 test,
 *                              32 the random stream's generation kernel (a wave of its pipeline gave up waiting; dasp_mt_randn checks too)
 *   dasp_device_error_clear()  re-arm
 *   dasp_plan_lookback(on)     1 (default): the one-launch look-back forms where they apply; 0: the two-launch forms (pre-pass + pass) -
 *                              no workgroup ever waits for another; < 0: query. Returns the setting.
 * Test support (tests/test_gpu_lookback.py): dasp_test_lookback_timeout(ms) sets the time-out (0 = 2 s); dasp_test_lookback_stall polls a
 * word that never arrives (zero_word: 8 zero bytes on the device; out: 1 float or NULL); dasp_test_spin keeps `workgroups` x `threads`
 * busy for `milliseconds`; dasp_test_stream_with_cus makes a stream confined to the first n_cus compute units (hipExtStreamCreateWithCUMask).
 * ------------------------------------------------------------------------------------------- */
int  dasp_device_error(void);
void dasp_device_error_clear(void);
int  dasp_plan_lookback(int on);
int  dasp_test_lookback_timeout(int milliseconds);
int  dasp_test_lookback_stall(const unsigned long long* zero_word, float* out, void* stream);
int  dasp_test_spin(int workgroups, int threads, double milliseconds, void* stream);
int  dasp_test_stream_with_cus(int n_cus, void** stream);
int  dasp_test_stream_destroy(void* stream);

/* ---------------------------------------------------------------------------------------------
 * Cascaded biquads.  Replaces dasp_pytorch.signal.sosfilt_via_fsm (dasp_pytorch/signal.py:136-166,
 * with fft_sosfreqz :14-32, fft_freqz :7-11, freqdomain_fir :35-39) and, through
 * dasp_peq_prepare, the coefficient design dasp_pytorch.signal.biquad (signal.py:242-306) as used
 * by dasp_pytorch.functional.parametric_eq (dasp_pytorch/functional.py:118-272).
 *
 * x, y, gy, gx: (B, C, N) fp32 contiguous; one filter per batch item shared by its C channels
 * (signal.py:157-158). Bs = number of filter sets: B, or 1 = broadcast over the batch
 * (functional.py:208-220). S = sections per filter; compiled for S in {2,4,6,8}.
 * ------------------------------------------------------------------------------------------- */
int  dasp_sos_supported_sections(int S);          /* 1 if a kernel for S sections is compiled */
int  dasp_sos_chunk(void);                        /* samples per lane chunk (L) */
int  dasp_sos_tile(void);                         /* samples per wave tile (64 L) */
int  dasp_sos_bwd_waves(void);                    /* waves per row in the backward kernel */
long dasp_sos_table_floats(int S);                /* fp32 table size per filter set */
long dasp_sos_dtab_doubles(int S);                /* fp64 side-table size per filter set */
long dasp_sos_num_tiles(long N);
long dasp_sos_carry_floats(long rows, long N, int S);   /* rows = B*C */
long dasp_sos_partial_floats(long rows, int S);          /* scratch between the backward kernel and the finalize step (16-byte aligned; opaque:
                                                          * one 32 x 32 fp64 Gram matrix per row or per (row, segment)) */

/* sos: (Bs, S, 6) fp32, rows [b0 b1 b2 a0 a1 a2] (signal.py:141). Fills tab / dtab. */
int dasp_sos_prepare(const float* sos, int Bs, int S, float* tab, double* dtab, void* stream);

/* params: (Bs, S, 3) fp32 rows [gain_db, cutoff_freq, q_factor]; types[S] (host array):
 * 0 peaking, 1 low_shelf, 2 high_shelf, 3 low_pass, 4 high_pass (signal.py:261-297). */
int dasp_peq_prepare(const float* params, int Bs, int S, const int* types, double sample_rate,
                     float* tab, double* dtab, void* stream);
/* The same from 3*S separate control vectors: rows = host array of device pointers, rows[3*k + c] -> Bs values of control c
 * (0 gain_db, 1 cutoff_freq, 2 q_factor) of section k - the tensors functional.parametric_eq receives (functional.py:118-139). */
int dasp_peq_prepare_rows(const float* const* rows, int Bs, int S, const int* types, double sample_rate, float* tab,
                          double* dtab, void* stream);

/* y = cascade(x). carries (may be NULL when no backward follows) receives the state of every lane chunk
 * (dasp_sos_carry_floats(rows, N, S) floats = 2*S per dasp_sos_chunk() samples, in the kernels' own order); the backward pass reads it
 * instead of re-scanning the forward recurrence. */
int dasp_sosfilt_forward(const float* tab, int Bs, const float* x, float* y, float* carries,
                         int B, int C, long N, int S, void* stream);

/* gx = adjoint cascade(gy); partials receives what dasp_sos_grad_finalize turns into the coefficient gradients (per row: the matrix
 * sum over chunks of [gy chunk; adjoint state] [x chunk; forward state]^T, accumulated on the matrix cores: csrc/sosfilt.hip sos_bwd_gram_kernel). */
int dasp_sosfilt_backward(const float* tab, int Bs, const float* x, const float* gy,
                          const float* carries, float* gx, float* partials,
                          int B, int C, long N, int S, void* stream);

/* mode 0: gout (B, S, 6) = dL/dsos;  mode 1: gout (B, S, 3) = dL/d[gain_db, cutoff_freq, q_factor].
 * With Bs == 1 the caller sums gout over the batch. */
int dasp_sos_grad_finalize(const double* dtab, int Bs, const float* partials, int B, int C, int S,
                           int mode, float* gout, void* stream);

/* dasp_sosfilt_backward followed by dasp_sos_grad_finalize (same mode / gout) as one call - the autograd of
 * signal.sosfilt_via_fsm (dasp_pytorch/signal.py:136-166) / functional.parametric_eq (functional.py:118-272) in one step.
 * Two launches at one workgroup per row (the segmented forms - dasp_peq_backward - finalize inside their launch and count an item's
 * workgroups in the item's table: hence the non-const tab; the count is left at zero). */
int dasp_sosfilt_backward_grads(float* tab, const double* dtab, int Bs, const float* x, const float* gy,
                                const float* carries, float* gx, float* partials, int mode, float* gout,
                                int B, int C, long N, int S, void* stream);

/* The backward pass as asked for (torch.autograd's needs_input_grad) and for the cascade it is:
 *   gx == NULL        no input gradient: the adjoint output is neither transposed back nor stored (parametric_eq is the first effect of
 *                     the reference's chain, examples/style_transfer.py:150 - its input never needs one);
 *   partials == NULL  no coefficient gradients (a fixed filter): x and carries are not read, only the adjoint cascade runs.
 * (Rounds 3 - 5 carried an `int designed` in these signatures that had been ignored since the Gram-matrix backward; round 6 dropped it -
 * the ABI hash changed with it.)
 * dasp_sos_grad_finalize_ex: segments = rows of partial sums per (row, wave) (1 after dasp_sosfilt_backward*, dasp_sos_segments(N, Tseg)
 * after the *_seg calls); mode 2 = mode 1 written as 3*S rows of B values ([3 k + c][item]: one contiguous vector per control tensor). */
int dasp_sosfilt_backward_ex(const float* tab, int Bs, const float* x, const float* gy, const float* carries, float* gx,
                             float* partials, int B, int C, long N, int S, void* stream);
int dasp_sos_grad_finalize_ex(const double* dtab, int Bs, const float* partials, int B, int C, int S, int segments, int mode,
                              float* gout, void* stream);
int dasp_sosfilt_backward_grads_ex(float* tab, const double* dtab, int Bs, const float* x, const float* gy, const float* carries,
                                   float* gx, float* partials, int mode, float* gout, int B, int C, long N, int S, void* stream);

/* functional.parametric_eq (dasp_pytorch/functional.py:118-272) as one call per direction. Forward = dasp_peq_prepare_rows +
 * dasp_sosfilt_forward (Tseg == 0) or + dasp_sos_segment_prepare + dasp_sosfilt_forward_seg (Tseg = dasp_sos_segment_tiles(B*C, N) > 0;
 * segtab / segbuf as below, NULL otherwise). Backward = the adjoint cascade and the control gradients from the tables the forward call
 * filled (designed cascade), gx / partials / mode as above; with Tseg > 0 partials hold dasp_sos_partial_floats(rows * segments, S). */
int dasp_peq_forward(const float* const* rows, int Bp, int S, const int* types, double sample_rate, float* tab, double* dtab,
                     const float* x, float* y, float* carries, int B, int C, long N, long Tseg, double* segtab, float* segbuf,
                     void* stream);
/* The same from the normalised (Bp, 3 S) tensor of Processor.process_normalized (dasp_pytorch/modules.py:25-91) - SURVEY 8(f1)'s fused op:
 * de-normalisation (lo, span: host arrays of 3 S doubles, min and max - min of every column, modules.py:136-155), range check (flag: one
 * device word, zeroed by the caller; bit i is set when column i leaves [0, 1] - read it back to raise the reference's ValueError,
 * modules.py:83-84; NULL = no check), design and cascade. dasp_peq_backward with mode 1 returns the gradient w.r.t. the normalised tensor. */
int dasp_peq_forward_norm(const float* pn, int Bp, int S, const int* types, double sample_rate, const double* lo, const double* span,
                          unsigned* flag, float* tab, double* dtab, const float* x, float* y, float* carries, int B, int C, long N,
                          long Tseg, double* segtab, float* segbuf, void* stream);
int dasp_peq_backward(float* tab, const double* dtab, int Bp, const float* x, const float* gy, const float* carries, float* gx,
                      float* partials, int mode, float* gout, int B, int C, long N, int S, long Tseg, const double* segtab,
                      float* segbuf, void* stream);
/* The design step of dasp_peq_forward_norm on its own: tables from the normalised (Bp, 3 S) tensor, no cascade. */
int dasp_peq_prepare_norm(const float* pn, int Bp, int S, const int* types, double sample_rate, const double* lo, const double* span,
                          unsigned* flag, float* tab, double* dtab, void* stream);
/* ... and, with Tseg > 0 (a power of two) and segtab, also the segment transition matrices of dasp_sos_segment_prepare, from the same launch. */
int dasp_peq_prepare_norm_seg(const float* pn, int Bp, int S, const int* types, double sample_rate, const double* lo, const double* span,
                              unsigned* flag, float* tab, double* dtab, long Tseg, double* segtab, void* stream);
/* The first two launches of dasp_sosfilt_forward_seg on their own (scan-only pre-pass + chain): afterwards the second half of segbuf
 * (dasp_sos_seg_floats floats) holds the state every (row, segment) starts from, [row][segment][2 S]. */
int dasp_sos_segment_starts(const float* tab, const double* segtab, int Bs, const float* x, float* segbuf, int B, int C, long N, int S,
                            long Tseg, void* stream);

/* ---- fused forward of the reference's effect chain: parametric EQ -> compressor -----------------------------------------------------------
 * examples/style_transfer.py:150-154 (equalizer -> compressor -> reverb -> gain) and :293-299 (the same chain without gradients, every
 * training step, to synthesise the target). y = compressor(parametric_eq(x)) in ONE pass over x: a workgroup owns a batch item (both
 * channels: the compressor's side chain is their sum, functional.py:328), a tile of the EQ's output goes through the gain computer, the
 * one-pole smoothing scan and the output multiply before it leaves the chip. 8 B per channel-sample instead of 16 for the two separate
 * forward calls; forward only (no chunk states or carries are saved). The chain's final gain commutes with the reverb and is folded
 * into makeup_gain_db by the caller (dasp_chain_controls).
 *   tab  : the EQ's tables (dasp_peq_prepare / _rows / _norm; Bs = 1 or B items; S = 6)      ctl : (B, 5) as for dasp_dynamics_forward
 *   mode : 0 compressor, 1 expander; no look-ahead                                            x, y: (B, C, N), C = 1 or 2
 *   Tseg : 0 = one workgroup per item, or dasp_chain_segment_tiles(B, N) (few items: every item cut into segments of Tseg tiles) with
 *          segtab from dasp_sos_segment_prepare(dtab, Bs, S, Tseg, ...) and segbuf of dasp_chain_seg_floats(B, C, N, S, Tseg) floats */
long dasp_chain_segment_tiles(long B, long N);
long dasp_chain_seg_floats(long B, long C, long N, int S, long Tseg);
int dasp_chain_forward(const float* tab, int Bs, const float* x, const float* ctl, float* y, int B, int C, long N, int S, int mode,
                       double sample_rate, float eps, long Tseg, const double* segtab, float* segbuf, void* stream);
/* The same pass for the step that carries gradients: also writes the EQ's output yeq (B, C, N) - the input of dasp_dynamics_backward -, the
 * EQ's chunk start states eq_carries (dasp_sos_carry_floats(B * C, N, S) floats, for dasp_peq_backward / dasp_sosfilt_backward*) and the
 * smoothing state entering every compressor tile dyn_carries (dasp_dyn_carry_floats(B, N) floats): 15 B per channel-sample instead of 19
 * for the two forward calls. One workgroup per item (no segments); pays from ~200 items on (profiles/r06/chain_fwd_saving_ab.log). */
int dasp_chain_forward_saving(const float* tab, int Bs, const float* x, const float* ctl, float* y, float* yeq, float* eq_carries,
                              float* dyn_carries, int B, int C, long N, int S, int mode, double sample_rate, float eps, void* stream);

/* dasp_pytorch.signal.biquad (dasp_pytorch/signal.py:242-306) as a call of its own: the fp64 RBJ design the prepare calls run, for n
 * (gain_db, cutoff_freq, q_factor) triples of one filter type. ba: (n, 6) fp64 rows [b0 b1 b2 1 a1 a2] (normalised by a0, as the
 * reference returns them); jac: (n, 15) fp64 = d(b0 b1 b2 a1 a2)/d(gain_db, cutoff_freq, q_factor), which dasp_biquad_backward
 * contracts with gba (n, 6), the gradient w.r.t. the rows of ba, into gparams (n, 3). */
int dasp_biquad_design(const double* gain_db, const double* cutoff_freq, const double* q_factor, int n, int type,
                       double sample_rate, double* ba, double* jac, void* stream);
int dasp_biquad_backward(const double* jac, const double* gba, int n, double* gparams, void* stream);

/* Few rows (B*C <= 128; from there up to 256 rows the plain calls launch twice the waves per row instead): a row is one workgroup, so the calls above
 * would leave most of the chip idle. The *_seg entry points cut every row into segments of Tseg tiles that run as independent workgroups;
 * same results (oracle/chunkscan_model.py forward_row_segmented / backward_row_segmented).
 *   dasp_sosfilt_forward_seg / dasp_sosfilt_backward_seg[_ex] (tables that may serve many calls): per direction a scan-only pre-pass gives
 *     every segment's end state, the last of an item's workgroups to finish chains them through Phi^(samples per segment)
 *     (dasp_sos_segment_prepare, from dtab; the completion counters are words of the item's table, zeroed by the prepare call and reset
 *     after use - the one place where a call writes into `tab`), then the ordinary pass runs per segment from its start state.
 *   dasp_peq_forward* / dasp_peq_backward (a design launch per call): ONE launch per direction - every workgroup sweeps its segment
 *     scan-only, hands the end state on as tagged 64-bit words in segbuf (8-byte aligned) and takes its start state from the words of the
 *     row's other segments (a decoupled look-back; the tag is drawn by the call's design launch, so dasp_peq_backward needs tables filled
 *     by dasp_peq_forward* of the same step), then runs the pass; the backward launch also finalizes the control gradients. Three
 *     launches per step: design, forward, backward.
 *   Tseg   = dasp_sos_segment_tiles(rows, N): proposed tiles per segment (a power of two), 0 = use the plain calls
 *   segtab = dasp_sos_segtab_doubles(S) doubles per item (Bs items): the two segment transition matrices + the basis responses of the Gram
 *            finalize step (filled by the design launch of dasp_peq_forward*);  segbuf = dasp_sos_seg_floats(rows, N, S, Tseg) floats of scratch
 *   partials of the backward pass: dasp_sos_partial_floats(rows * dasp_sos_segments(N, Tseg), S) floats, finalized by
 *   dasp_sos_grad_finalize_seg(..., segments = dasp_sos_segments(N, Tseg), ...) after dasp_sosfilt_backward_seg*. */
long dasp_sos_segment_tiles(long rows, long N);
long dasp_sos_segments(long N, long Tseg);
long dasp_sos_segtab_doubles(int S);
long dasp_sos_seg_floats(long rows, long N, int S, long Tseg);
int dasp_sos_segment_prepare(const double* dtab, int Bs, int S, long Tseg, double* segtab, void* stream);
int dasp_sosfilt_forward_seg(const float* tab, const double* segtab, int Bs, const float* x, float* y, float* carries,
                             float* segbuf, int B, int C, long N, int S, long Tseg, void* stream);
int dasp_sosfilt_backward_seg(const float* tab, const double* segtab, int Bs, const float* x, const float* gy,
                              const float* carries, float* gx, float* partials, float* segbuf,
                              int B, int C, long N, int S, long Tseg, void* stream);
int dasp_sosfilt_backward_seg_ex(const float* tab, const double* segtab, int Bs, const float* x, const float* gy,
                                 const float* carries, float* gx, float* partials, float* segbuf,
                                 int B, int C, long N, int S, long Tseg, void* stream);
int dasp_sos_grad_finalize_seg(const double* dtab, int Bs, const float* partials, int B, int C, int S, int segments,
                               int mode, float* gout, void* stream);

/* ---------------------------------------------------------------------------------------------
 * gain / distortion.  Replace dasp_pytorch.functional.gain (dasp_pytorch/functional.py:10-29):
 * y = x * 10^(gain_db/20), gain_db (B) one value per batch item repeated over channels (:26-28);
 * and dasp_pytorch.functional.distortion (functional.py:65-78): y = tanh(x * 10^(drive_db/20)),
 * drive_db (B*C) one value per (b, c) row (the reference's drive_db.view(bs, chs, -1), :78).
 * Backward: gx = dL/dx, ggain (B) / gdrive (B*C) = dL/d(control in dB); `partials` is scratch of
 * dasp_ew_partial_floats(B*C, N) floats.
 * ------------------------------------------------------------------------------------------- */
long dasp_ew_partial_floats(long rows, long N);
int dasp_gain_forward(const float* x, const float* gain_db, float* y, int B, int C, long N, void* stream);
int dasp_gain_backward(const float* x, const float* gain_db, const float* gy, float* gx, float* ggain,
                       float* partials, int B, int C, long N, void* stream);
int dasp_distortion_forward(const float* x, const float* drive_db, float* y, int B, int C, long N, void* stream);
int dasp_distortion_backward(const float* x, const float* drive_db, const float* gy, float* gx, float* gdrive,
                             float* partials, int B, int C, long N, void* stream);
/* distortion with one drive value per SAMPLE: the reference's drive_db.view(bs, chs, -1) also takes bs * chs * seq_len values
 * (functional.py:78). x, drive_db, y, gy, gx, gdrive: n = B * C * N floats each; gdrive = dL/d(drive_db) per sample. */
int dasp_distortion_sample_forward(const float* x, const float* drive_db, float* y, long n, void* stream);
int dasp_distortion_sample_backward(const float* x, const float* drive_db, const float* gy, float* gx, float* gdrive, long n, void* stream);

/* Controls of the reference's effect chain (examples/style_transfer.py:150-154: equalizer -> compressor -> reverb -> gain, each through
 * Processor.process_normalized, dasp_pytorch/modules.py:25-51) from the normalised parameter tensors in one launch: comp_pn (B, 6),
 * reverb_pn (B, 25), gain_pn (B, 1) in [0, 1]; lo, span: HOST arrays of 32 floats, [0, 6) the compressor's ranges in the order of its
 * param_ranges (modules.py:159-187), [6, 31) the reverb's, [31] the gain's. Out: ctl (B, 5) as dasp_dynamics_forward takes it, with the
 * gain added to the make-up gain (a per-item gain commutes with the linear reverb); gains, decays (B, 12), mix (B) as
 * dasp_reverb_forward takes them. The backward call maps the gradients of those four back to the three parameter tensors. */
/* flag (may be NULL): one device word; bit i is OR-ed in when column i of the 32 (compressor 0-5, reverb 6-30, gain 31) holds a value outside
 * [0, 1] (the reference's ValueError, modules.py:83-84: the host reads the word back when it chooses to; it is never cleared here). */
int dasp_chain_controls(const float* comp_pn, const float* reverb_pn, const float* gain_pn, const float* lo, const float* span, float* ctl,
                        float* gains, float* decays, float* mix, unsigned* flag, int B, void* stream);
int dasp_chain_controls_backward(const float* gctl, const float* ggain, const float* gdecay, const float* gmix, const float* span,
                                 float* gcomp_pn, float* greverb_pn, float* ggain_pn, int B, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dynamics: compressor / expander.  mode 0 replaces dasp_pytorch.functional.compressor
 * (dasp_pytorch/functional.py:275-399: side-chain sum :328, level in dB :347, soft-knee gain
 * computer :350-369, one-pole smoothing :372-380 which the reference evaluates through
 * signal.lfilter_via_fsm, dasp_pytorch/signal.py:95-133, look-ahead delay :383-385, make-up and
 * dB->linear :388-394). mode 1 is a downward expander of the same structure; the reference's
 * expander is a stub (functional.py:402-403), so it has no reference behaviour.
 *
 * x, y, gy, gx: (B, C, N) fp32 contiguous. ctl: (B, 5) fp32 rows [threshold_db, ratio, attack_ms,
 * knee_db, makeup_gain_db] (release_ms is unused by the reference, :340,343-344, and has no entry).
 * carries: dasp_dyn_carry_floats(B, N) floats written by forward, read by backward (may be NULL in
 * forward when no backward follows). lin_buf: (B, N) floats, required only when lookahead > 0
 * (forward writes the linear gain, backward reads it at shifted positions). gctl: (B, 5) = dL/dctl.
 * partials: scratch of dasp_dyn_partial_floats(B) floats.
 * ------------------------------------------------------------------------------------------- */
long dasp_dyn_num_tiles(long N);
long dasp_dyn_carry_floats(long B, long N);
long dasp_dyn_partial_floats(long B);
int dasp_dyn_counters_reset(int* counters, int B, void* stream);
int dasp_dynamics_forward(int mode, const float* x, const float* ctl, float* y, float* carries, float* lin_buf,
                          int B, int C, long N, double sample_rate, float eps, int lookahead, void* stream);
int dasp_dynamics_backward(int mode, const float* x, const float* ctl, const float* gy, const float* carries,
                           const float* lin_buf, float* gx, float* gctl, float* partials, int B, int C, long N,
                           double sample_rate, float eps, int lookahead, void* stream);

/* Few batch items (B < 128; the reference trains with 8 to 32: examples/style_transfer.py:403, auto_eq.py:231): one workgroup per item
 * would leave most of the chip idle, so every item is cut into segments of Tseg tiles that run as independent workgroups - scan-only
 * pre-pass, a scalar chain through alpha^(samples per segment) in fp64, then the ordinary pass per segment; same results.
 *   Tseg = dasp_dyn_segment_tiles(B, N) (a power of two; 0 = use the plain calls); segbuf = 2 * B * dasp_dyn_segments(N, Tseg) floats of
 *   scratch; carries as above; partials of the backward pass: dasp_dyn_partial_floats(B * dasp_dyn_segments(N, Tseg)) floats. */
long dasp_dyn_segment_tiles(long B, long N);
long dasp_dyn_segments(long N, long Tseg);
int dasp_dynamics_forward_seg(int mode, const float* x, const float* ctl, float* y, float* carries, float* lin_buf, float* segbuf,
                              int B, int C, long N, double sample_rate, float eps, int lookahead, long Tseg, int* counters, void* stream);
int dasp_dynamics_backward_seg(int mode, const float* x, const float* ctl, const float* gy, const float* carries,
                               const float* lin_buf, float* gx, float* gctl, float* partials, float* segbuf, int B, int C, long N,
                               double sample_rate, float eps, int lookahead, long Tseg, int* counters, void* stream);
/* functional.compressor / expander on the reference's own six control tensors (dasp_pytorch/functional.py:275-286), without a stacking
 * launch in front of the kernels or a transposition behind them: rows = 5 device vectors of B floats (threshold_db, ratio, attack_ms,
 * knee_db, makeup_gain_db; the array of pointers itself is host memory), grows = 6 device vectors of B floats that receive the gradients
 * in the reference's argument order (threshold_db, ratio, attack_ms, release_ms - set to zero: it has no path to the output,
 * functional.py:340,343-344 -, knee_db, makeup_gain_db). Tseg = 0: one workgroup per item (segbuf / counters unused), otherwise as the
 * *_seg calls above. Everything else as dasp_dynamics_forward / _backward. */
int dasp_dynamics_forward_rows(int mode, const float* x, const float* const* rows, float* y, float* carries, float* lin_buf, float* segbuf,
                               int B, int C, long N, double sample_rate, float eps, int lookahead, long Tseg, int* counters, void* stream);
int dasp_dynamics_backward_rows(int mode, const float* x, const float* const* rows, const float* gy, const float* carries,
                                const float* lin_buf, float* gx, float* const* grows, float* partials, float* segbuf, int B, int C, long N,
                                double sample_rate, float eps, int lookahead, long Tseg, int* counters, void* stream);
/* counters (may be NULL): a buffer of AT LEAST 4 * B ints owned by the caller, one buffer per stream, ZERO before its first use; every
 * call returns the words it used to zero. dasp_dyn_counters_reset(counters, B, stream) zeroes them on the stream (a kernel): call it after
 * allocating the buffer and after any segmented call that returned an error - a launch that did not complete may leave a count behind,
 * and a count that never completes means wrong start states and gradients for every later call (both bindings of this repo drop their
 * cached buffer when a call fails). (Rounds 3 - 4 zeroed them with hipMemsetAsync at the start of every call; inside a captured
 * graph that memset node was not ordered before the kernel behind it when a replay started on an idle device, so the library zeroes
 * nothing with memset nodes any more - csrc/common.hpp zero_async.) With them the forward pass is ONE launch when Tseg is 1, 2 or 4
 * tiles per forward wave (16, 32, 64: every workgroup runs its segment from a zero state, hands the segment's end state on as a tagged
 * 64-bit word in segbuf - 8-byte aligned - and corrects its tile carries by the look-back sum over the item's earlier segments; the
 * smoothing state is one float that enters linearly) and the backward pass is one launch when Tseg is two tiles per backward wave (16)
 * with lookahead 0, C <= 2 and 16-byte aligned rows; otherwise the item's last workgroup of a pre-pass chains the segments and the last
 * one of the adjoint pass forms the control gradients: two launches per direction instead of three / four (workgroup to workgroup by
 * agent-scope atomics, csrc/common.hpp handoff_arrive_is_last, as for the biquad cascade). */

/* ---------------------------------------------------------------------------------------------
 * Noise-shaped reverberation.  Replaces dasp_pytorch.functional.noise_shaped_reverberation
 * (dasp_pytorch/functional.py:406-577): grouped FIR filter bank over white noise (:548-558),
 * per-band exponential envelope x gain and mean over bands (:561-567), causal convolution of the
 * input with the resulting 2-channel impulse response truncated to N (:570-572), wet/dry mix (:575).
 * The filter design (dasp_pytorch.signal.octave_band_filterbank, dasp_pytorch/signal.py:42-92,
 * SciPy firwin on the host) stays on the host; its taps are passed in.
 *
 * Both convolutions run on the library's own register/LDS FFTs (no FFT library is linked or loaded): the filter
 * bank (taps <= 3585) as one fused kernel that is re-run in the backward pass instead of saving its output, the
 * long convolution (L <= 2^20) as overlap-add with four-step transforms of pairs of blocks (reverb.hip).
 *
 * x, y, gy, gx: (B, 2, N) fp32.  noise: (2B, nb, L + taps - 1).  gains, decays: (B, nb).  mix: (B).
 * nb <= 16 bands, L = IR length, taps = FIR length.  All buffer sizes come from dasp_reverb_sizes.
 * ------------------------------------------------------------------------------------------- */
/* Plan overrides of the reverb - explicit arguments for the three choices its planner makes (rounds 2 - 4 read them from the environment;
 * a library should not): chunk = signals per pass of the long-convolution pipeline (<= 0: all at once, the default), weight_limit = largest
 * |rho| * 4096 served with the envelope inside the filter bank's transform (0: every item the per-band way; < 0: the default),
 * band_split = workgroups per (item, window) of the filter bank (< 1: the planner's rule). Process-wide: set before dasp_reverb_sizes and
 * the calls that use its numbers, set back to (-1, -1, -1) afterwards. Developer A/Bs and the tests of the alternative routes. */
int dasp_reverb_plan(long chunk, float weight_limit, int band_split);

/* sizes[0] = block length Lb, [1] = transform length n1, [2] = pairs of blocks per signal, [3] = blocks per signal,
 * [4] = complex (2 x fp32) elements of Fspec, [5] = filter-bank windows per batch item,
 * [6] = complex elements of A (2B * pairs * n1), [7] = complex elements of H (B * n1: the two impulse responses of an item are one complex
 *       frame, left + i right), [8] = floats of each of ir / gir,
 * [9] = signals per pass of the long-convolution pipeline (all 2B by default; a developer switch can cut the pipeline into passes over
 *       chunks of signals that reuse chunk-sized scratch buffers),
 * [10] = floats of mix_part, [11] = floats of part,
 * [12] = complex elements of each of the scratch buffers W / W2 / Ag (at least B * nb * 4096: W and Ag also hold the per-item weighted
 *        band spectra of the filter bank while it runs), [13] = complex elements of each of the scratch buffers Ah / P. */
int dasp_reverb_sizes(int B, long N, int L, int taps, int nb, long* sizes /* [14] */);
/* filters (nb, taps) fp32, the host-designed bank -> Fspec: transform twiddles, the band spectra, the taps themselves (the filter bank
 * re-weights them by each item's decay envelope per call). */
int dasp_reverb_filter_spectrum(const float* filters, int nb, int taps, void* Fspec, void* stream);
/* forward: H, ir (the impulse responses) and (when a backward pass follows) A are kept for it - pass A = NULL otherwise and the column
 * transforms of x go to the scratch W2; W, Ah are scratch.
 * backward: takes the forward call's ir (first argument), H and A - not x; gx, ggain (B, nb), gdecay (B, nb), gmix (B) are the results;
 * Ag, W, P, gir, part, mix_part are scratch. Neither the wet signal nor x is needed for d loss / d mix = sum_n gy (wet - x):
 * sum_n gy wet = sum_r ir[r] c[r] and sum_n gy x = c[0] with c[r] = sum_n gy[n] x[n - r], r < L - the correlation the pass forms for
 * d loss / d ir anyway. */
int dasp_reverb_forward(const float* x, const float* noise, const void* Fspec, const float* gains, const float* decays,
                        const float* mix, float* y, void* A, void* H, void* W, void* W2, void* Ah, float* ir, int B, int Cx,
                        long N, int L, int taps, int nb, float decay_bound, void* stream);
int dasp_reverb_backward(const float* ir, const float* gy, const float* noise, const void* Fspec, const float* gains,
                         const float* decays, const float* mix, const void* A, const void* H, float* gx,
                         float* ggain, float* gdecay, float* gmix, void* Ag, void* W, void* P, float* gir, float* part,
                         float* mix_part, int B, int Cx, long N, int L, int taps, int nb, float decay_bound, void* stream);
/* Cx: channels of x, 2, or 1 for a mono input (the reference duplicates it to stereo, functional.py:493-495: here both output channels read
 * the one row - no copy; gx stays (B, 2, N), the gradient w.r.t. the mono input is the sum of its two rows).
 * decay_bound: > 0 = the caller vouches that no band decay exceeds this value (Processor.process_normalized: the validated upper end of the
 * parameter range); when even that decay stays on the envelope-inside-the-transform route of the filter bank, the other route's (empty) launch
 * is skipped. 0 = unknown.
 * The same two calls with the white noise of functional.py:548 generated inside the filter-bank kernels from a 64-bit seed instead of read
 * from memory (the `device_noise=True` mode of the Python layer): a counter-based stream, a pure function of (seed, batch item, band,
 * sample index) - one 32-bit hash per (item, band, index), its halves a Box-Muller pair = the item's two noise rows - so the forward and
 * the backward pass recompute identical values and nothing of size (2B, nb, L + taps - 1) exists (0.82 GB at B = 128 and the default
 * sizes, which torch.randn wrote once and the two filter-bank kernels read once each). Both calls of a step take the same seed.
 * seed_dev (may be NULL): one 64-bit device word added to `seed` when the kernels run - a launch captured into a HIP graph has its by-value
 * seed frozen; bumping the word between replays gives every replay new noise.
 * dasp_reverb_noise writes the stream out in the reference's layout, out (2B, nb, row_len), row_len = L + taps - 1 < 2^24: a test hook
 * (the explicit-noise calls above, given that tensor, must reproduce the seeded calls). Specification: oracle/noise_stream.py. */
int dasp_reverb_forward_rng(const float* x, unsigned long long seed, const unsigned long long* seed_dev, const void* Fspec, const float* gains, const float* decays,
                            const float* mix, float* y, void* A, void* H, void* W, void* W2, void* Ah, float* ir, int B, int Cx,
                            long N, int L, int taps, int nb, float decay_bound, void* stream);
int dasp_reverb_backward_rng(const float* ir, const float* gy, unsigned long long seed, const unsigned long long* seed_dev, const void* Fspec, const float* gains,
                             const float* decays, const float* mix, const void* A, const void* H, float* gx,
                             float* ggain, float* gdecay, float* gmix, void* Ag, void* W, void* P, float* gir, float* part,
                             float* mix_part, int B, int Cx, long N, int L, int taps, int nb, float decay_bound, void* stream);
int dasp_reverb_noise(unsigned long long seed, const unsigned long long* seed_dev, float* out, int B, int nb, long row_len, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stereo utilities.  Replace dasp_pytorch.functional.stereo_widener (dasp_pytorch/functional.py:580-605),
 * stereo_panner (:608-636) and stereo_bus (:32-62).
 *   widener: x, y (B, 2, N); width (B).            left = L + (1 - 2 width) R, right = (1 - 2 width) L + R
 *   panner:  x (B, T, N); pan (B * T); y (B, 2, T, N).  y[:, 0] = lg x, y[:, 1] = rg x, constant-power-like gains of pan
 *   bus:     x (B, 2, T, N); send_db (B * T); y (B, 2, N).  y = sum over tracks of 10^(send_db / 20) x;  T <= 64
 * partials: dasp_stereo_partial_floats(op, B, T, N) floats (op 0 widener, 1 panner, 2 bus).
 * ------------------------------------------------------------------------------------------- */
long dasp_stereo_partial_floats(int op, long B, int T, long N);
int dasp_widener_forward(const float* x, const float* width, float* y, int B, long N, void* stream);
int dasp_widener_backward(const float* x, const float* width, const float* gy, float* gx, float* gwidth, float* partials, int B,
                          long N, void* stream);
int dasp_panner_forward(const float* x, const float* pan, float* y, int B, int T, long N, void* stream);
int dasp_panner_backward(const float* x, const float* pan, const float* gy, float* gx, float* gpan, float* partials, int B, int T,
                         long N, void* stream);
int dasp_bus_forward(const float* x, const float* send_db, float* y, int B, int T, long N, void* stream);
int dasp_bus_backward(const float* x, const float* send_db, const float* gy, float* gx, float* gsend, float* partials, int B, int T,
                      long N, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-resolution STFT loss, the op downstream of the effect chain in the reference's training loops
 * (auraloss.freq.MultiResolutionSTFTLoss(): examples/style_transfer.py:341,363, auto_eq.py:252, virtual_analog.py:288; auraloss is
 * not vendored by the reference - its published algorithm is restated in oracle/dasp_oracle.py:mrstft_loss).
 * pred, target: (rows, N) fp32.  Resolutions: nres <= 8 triples (fft, hop, win); fft a power of two in 8..4096, win <= fft, fft/2 < N.
 * tw: 4096 complex twiddles from dasp_mrstft_table.  partials: dasp_mrstft_partial_floats floats.  stats: 4*nres floats
 * (forward -> backward).  loss, gloss: device scalars.  gpred = gloss * d loss / d pred (float atomics: summation order only
 * is not deterministic).
 * ------------------------------------------------------------------------------------------- */
long dasp_mrstft_partial_floats(long rows, int N, int nres, const int* fft, const int* hop, const int* win);
int dasp_mrstft_table(void* tw, void* stream);
int dasp_mrstft_forward(const float* pred, const float* target, const void* tw, float* partials, float* stats, float* loss, int rows,
                        int N, int nres, const int* fft, const int* hop, const int* win, float eps, void* stream);
int dasp_mrstft_backward(const float* pred, const float* target, const void* tw, const float* stats, const float* gloss, float* gpred,
                         int rows, int N, int nres, const int* fft, const int* hop, const int* win, float eps, void* stream);
/* the gradient w.r.t. the second signal (auraloss differentiates both): gtarget (rows, N) = gloss * d loss / d target */
int dasp_mrstft_backward_target(const float* pred, const float* target, const void* tw, const float* stats, const float* gloss,
                                float* gtarget, int rows, int N, int nres, const int* fft, const int* hop, const int* win, float eps,
                                void* stream);

/* ---------------------------------------------------------------------------------------------
 * Filters longer than one biquad.  Replaces dasp_pytorch.signal.lfilter_via_fsm (dasp_pytorch/signal.py:95-133) for K = 4 .. 16
 * coefficients per row (orders 3 .. 15; K <= 3 goes through the cascaded-biquad entry points above; the reference's only caller uses
 * K = 2, functional.py:372-380). Exact recurrence in double arithmetic, one thread per (row, chunk of time); the chunks run side by side
 * and are stitched by their state transition (csrc/lfilter.hip: three launches per direction).
 *   x, y, gy, gx: (rows, N) float (f64 = 0) or double (f64 = 1);  bn, an: (Bs, K) doubles, Bs = rows or 1, normalised so that
 *   an[:, 0] = 1 (FIR: an = 1, 0, ...);  wsave: (N, rows) doubles written by the forward pass for the backward pass (NULL: none follows);
 *   work: dasp_lfilter_work_doubles(rows, N, K, chunk) doubles of scratch, its size passed as work_doubles (checked: DASP_ERR_ARG);
 *   chunk: samples per chunk of time, 0 = the library's plan (1024 from 16 chunks on). The size query, the forward and the backward call of
 *   one filter operation must be given the same value; the library reads no environment variable for it (round 4, advisor finding);
 *   gb, ga: (rows, K) doubles, per row (the caller adds the rows of a broadcast filter; ga[:, 0] = 0);  gx may be NULL.
 * ------------------------------------------------------------------------------------------- */
long dasp_lfilter_work_doubles(int rows, long N, int K, long chunk);
int dasp_lfilter_forward(const void* x, const double* bn, const double* an, int Bs, void* y, double* wsave, double* work, long work_doubles,
                         int rows, long N, int K, int f64, long chunk, void* stream);
int dasp_lfilter_backward(const void* gy, const double* bn, const double* an, int Bs, const double* wsave, void* gx, double* gb,
                          double* ga, double* work, long work_doubles, int rows, long N, int K, int f64, long chunk, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Double precision.  The reference follows the dtype of its input (`.type_as(x)`, dasp_pytorch/signal.py:113,119,
 * functional.py:211), so float64 tensors mean float64 arithmetic. These entry points are that path for the recurrences and the
 * elementwise effects - the same maps as above evaluated plainly, one thread per row / batch item, sequential in time: meant for
 * validation, torch.autograd.gradcheck and small reference runs, not for throughput (csrc/ref64.hip).
 *   c5: (Bs, S, 5) fp64 normalised coefficients [b0 b1 b2 a1 a2] (from dasp_sos64_normalize, or rows of dasp_biquad_design's ba);
 *   wsave: (B*C, S, N) doubles written by the forward pass for the coefficient gradients (NULL when none are wanted);
 *   gc5: (Bs, S, 5) gradient w.r.t. c5, mapped by dasp_sos64_grads to sos (mode 0) or to (gain_db, cutoff_freq, q_factor) (mode 1,
 *   jac = the (Bs, S, 15) Jacobians of dasp_biquad_design);  gsave: (B, N) smoothed gain kept by dasp_dynamics64_forward.
 * ------------------------------------------------------------------------------------------- */
int dasp_sos64_normalize(const double* sos, int Bs, int S, double* c5, void* stream);
int dasp_sos64_forward(const double* c5, int Bs, const double* x, double* y, double* wsave, int B, int C, long N, int S, void* stream);
int dasp_sos64_backward(const double* c5, int Bs, const double* gy, const double* wsave, double* gx, double* gc5, int B, int C, long N,
                        int S, void* stream);
int dasp_sos64_grads(const double* c5, const double* sos, const double* gc5, const double* jac, int Bs, int S, int mode, double* out,
                     void* stream);
int dasp_dynamics64_forward(int mode, const double* x, const double* ctl, double* y, double* gsave, int B, int C, long N,
                            double sample_rate, double eps, int lookahead, void* stream);
int dasp_dynamics64_backward(int mode, const double* x, const double* ctl, const double* gy, const double* gsave, double* gx,
                             double* gctl, int B, int C, long N, double sample_rate, double eps, int lookahead, void* stream);
/* op 0: gain (ctl: B values), op 1: distortion (ctl: B*C values) */
int dasp_ew64_forward(int op, const double* x, const double* ctl, double* y, int B, int C, long N, void* stream);
int dasp_ew64_backward(int op, const double* x, const double* ctl, const double* gy, double* gx, double* gctl, int B, int C, long N,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * The reverb's white noise as the reference draws it.  Replaces `torch.randn(bs*2, 12, num_samples + num_bandpass_taps - 1)`
 * on the global CPU generator (dasp_pytorch/functional.py:548) by the same stream generated on the device: at::mt19937 run from
 * the state the host hands over, one 24-bit float per word, torch's 16-wide Box-Muller layout with its tail rule
 * (csrc/mtrand.hip; host side - state parsing, jump-ahead table - dasp_pytorch_amd/_mt19937.py).
 *   state_host: the 624 words of the generator (host memory, read during the call);  left: its `left` field (1..624);
 *   out: n >= 16 floats (device);  table: the (262, row stride) uint16 jump table on the device, laid out as dasp_mt_layout says
 *   (out8 = {regenerations per unit, baby polynomials, giant polynomials, exponents per uint16 (bit s of word k of a row = the
 *   coefficient of t^(16 k + s)), row stride, window length, max units per call, 0});
 *   scratch: dasp_mt_scratch_words(left, n) 32-bit words (device); after the stream has run, words [*final_state_offset_words, + 624)
 *   hold the generator's words after the draw (when *regenerated != 0; else they are unchanged and not written) and *left_after its
 *   `left`. One call takes at most dasp_mt_max_values() values (a caller with more loops over pieces that are multiples of 16).
 * ------------------------------------------------------------------------------------------- */
int dasp_mt_layout(int* out8);
long long dasp_mt_max_values(void);
long dasp_mt_scratch_words(int left, long long n);
int dasp_mt_randn(const unsigned* state_host, int left, float* out, long long n, const unsigned short* table, unsigned* scratch,
                  int* left_after, int* regenerated, long* final_state_offset_words, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DASP_HIP_H */
