// torch.ops.dasp.*: the PyTorch-ROCm extension over the C ABI (include/dasp_hip.h). Two groups of ops:
//   * the reference's own callables (round 5) - parametric_eq on its 3 S control tensors (dasp_pytorch/functional.py:118-139), dynamics on
//     the six controls of compressor / expander (:275-286), gain (:10), distortion (:65), sosfilt (signal.py:136), noise_shaped_reverb on its
//     12 + 12 + 1 control tensors (functional.py:406-436): dasp_pytorch_amd.functional routes float32 ROCm tensors through them;
//   * the effect chain of the reference's training loop (examples/style_transfer.py:150-154) on normalised parameters (round 4):
//     parametric_eq_norm, dynamics_ctl, reverb on control matrices, and the chain's fused control de-normalisation chain_controls. What SURVEY 8(b) / BASELINE north_star specify: ops
// registered with TORCH_LIBRARY (schemas visible to torch.compile / torch.library.opcheck), forward + hand-derived adjoint as
// torch::autograd::Function in C++ (the backward pass runs on autograd's worker thread without the Python interpreter), errors as
// TORCH_CHECK -> RuntimeError. No kernels live here: every number comes from libdasp_hip.so, launched on torch's current HIP stream.
// The ctypes binding (dasp_pytorch_amd/_lib.py, ops.py) stays as the no-torch-extension path and for every other op.
//
// Layout of an op family (as torchvision's roi_align): a public differentiable op with a kernel on the dispatch key of the device
// (inference: nothing saved) and one on Autograd (a Function whose forward calls the `_forward` op - outputs + what the adjoint needs -
// and whose backward calls the `_backward` op). The `_forward` / `_backward` ops are ordinary functional ops with fake (meta)
// implementations registered from Python (dasp_pytorch_amd/_torch_ops.py), so AOTAutograd traces through both directions.
// Backward ops have no derivative formula: differentiating twice raises (the hand-written adjoints are once-differentiable).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/autograd.h>
#include <torch/csrc/autograd/autograd_not_implemented_fallback.h>
#include <torch/library.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include <hip/hip_runtime_api.h>

#include "dasp_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// The completion counters of the segmented compressor calls (dasp_hip.h: 4 ints per item, zero before their first use, every use returns
// them to zero): one buffer per (device, stream), as ops._dyn_counters. Inside a graph capture a fresh zeroed buffer from the graph's
// pool (a fill KERNEL node; the library no longer zeroes them with a memset node - csrc/dynamics.hip).
// (A call that fails may leave a count behind: check_rc drops the cached buffers, the next call starts from fresh zeros - advisor, r05.)
std::mutex g_counters_mu;
std::map<std::pair<int, void*>, Tensor> g_counters_cache;
Tensor dyn_counters(const Tensor& like, int64_t B) {
    const auto opts = like.options().dtype(at::kInt);
    hipStream_t st = (hipStream_t)stream_of(like);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone || B > 128) return at::zeros({4 * B}, opts);
    auto& mu = g_counters_mu;
    auto& cache = g_counters_cache;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair((int)like.device().index(), (void*)st);
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (cache.size() >= 64) cache.clear();
        it = cache.emplace(key, at::zeros({4 * 128}, opts)).first;
    }
    return it->second;
}

void check_rc(int rc, const char* what) {
    if (rc != 0) {
        std::lock_guard<std::mutex> lock(g_counters_mu);
        g_counters_cache.clear();
    }
    TORCH_CHECK(rc == 0, what, " failed: ", rc == -1 ? "DASP_ERR_ARG (bad argument)" : rc == -2 ? "DASP_ERR_UNSUPPORTED" : rc == -3 ?
                "DASP_ERR_DEVICE (a kernel of an EARLIER segmented call gave up waiting for a look-back word and wrote NaN: family bits of dasp_device_error(); "
                "dasp_pytorch_amd.config.plan.lookback = False selects the two-launch forms, dasp_device_error_clear() re-arms) " : "HIP error ", rc);
}
void need_device(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "dasp: `", name, "` must be on a ROCm device (there is no CPU path), got ", t.device());
}
void same_device(const Tensor& a, const Tensor& b, const char* name) {
    TORCH_CHECK(a.device() == b.device(), "dasp: `", name, "` is on ", b.device(), " but x is on ", a.device());
}
Tensor f32c(const Tensor& t) { return t.to(at::kFloat).contiguous(); }
// For the autograd functions (which run ABOVE the Python dispatch key, where a composite op that hands its input back unchanged - reshape of
// a flat tensor, to() of the right dtype - is seen as "an operator returning its input" by torch.library.opcheck's schema test): numel
// through the symbolic interface (AOT tracing with dynamic shapes), and a flat float32 view / copy that only calls what changes something.
int64_t nel(const Tensor& t) { return t.sym_numel().guard_int(__FILE__, __LINE__); }
// a gradient row in the shape and dtype of the control tensor it belongs to
Tensor like_control(const Tensor& g, const Tensor& c) {
    Tensor r = g.reshape_as(c);
    return r.scalar_type() == c.scalar_type() ? r : r.to(c.scalar_type());
}
Tensor flat32(const Tensor& c) {
    Tensor t = c;
    if (t.dim() != 1) t = t.reshape({-1});
    if (t.scalar_type() != at::kFloat) t = t.to(at::kFloat);
    return t;
}
float* fp(const Tensor& t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }
double* dp(const Tensor& t) { return t.defined() && t.numel() ? t.data_ptr<double>() : nullptr; }
long round64(long n) { return (n + 63) & ~63L; }
// the optional range-flag word of the normalised ops: one int32 on x's device into which the kernels OR a bit per parameter column that
// left [0, 1] (the reference's ValueError, modules.py:83-84; read back by dasp_pytorch_amd.modules when it chooses to - never cleared here)
unsigned* flag_ptr(const c10::optional<Tensor>& f, const Tensor& x) {
    if (!f.has_value() || !f->defined()) return nullptr;
    TORCH_CHECK(f->is_cuda() && f->device() == x.device() && f->scalar_type() == at::kInt && f->numel() >= 1 && f->is_contiguous(),
                "dasp: range_flag must be an int32 tensor on x's device");
    return reinterpret_cast<unsigned*>(f->data_ptr<int>());
}
Tensor empty_f32(long n, const Tensor& like) { return at::empty({n}, like.options().dtype(at::kFloat)); }

// ---- parametric EQ on the normalised (Bp, 3 S) tensor (ops.ParametricEQNormFunction; dasp_peq_forward_norm / dasp_peq_backward) ---------
// tseg: tiles per segment of the segmented-row kernels, 0 = one workgroup per row. Decided ONCE per op call (planner, or the developer
// overrides DASP_SOS_SEGMENT / DASP_SOS_SEGMENT_TILES) by the autograd function and handed to both directions.
// The planner's proposal, or the developer override dasp_pytorch_amd._torch_ops pushes in (dasp::_plan_override: -1 = the planner, 0 = never
// segment, > 0 = that many tiles per segment; the Python layer reads DASP_SOS_SEGMENT / DASP_SOS_SEGMENT_TILES / DASP_DYN_SEGMENT / _TILES -
// this file reads no environment).
int64_t g_sos_tiles_override = -1, g_dyn_tiles_override = -1;
void plan_override(int64_t sos_tiles, int64_t dyn_tiles) { g_sos_tiles_override = sos_tiles; g_dyn_tiles_override = dyn_tiles; }
// generic: a cascade given by its coefficients (sosfilt: no design launch per call, so its segmented rows keep the pre-pass launches, five
// launches per step) - there segments stop paying above 64 rows (profiles/r04/seg_crossover.log); the designed paths (three launches) go
// up to the planner's 128 (profiles/r05/seg_crossover.log)
int64_t sos_segment_tiles(int64_t rows, int64_t N, bool generic = false) {
    if (g_sos_tiles_override >= 0) return g_sos_tiles_override;
    return generic && rows > 64 ? 0 : dasp_sos_segment_tiles(rows, N);
}
struct PeqDims { int64_t B, C, N, Bp, S; };
PeqDims peq_check(const Tensor& x, const Tensor& pn, at::IntArrayRef types, at::ArrayRef<double> lo, at::ArrayRef<double> span) {
    need_device(x, "x");
    same_device(x, pn, "param_tensor");
    TORCH_CHECK(x.dim() == 3, "dasp::parametric_eq_norm: x must be (bs, chs, seq_len), got ", x.sizes());
    TORCH_CHECK(x.scalar_type() == at::kFloat, "dasp::parametric_eq_norm computes in float32; got ", x.scalar_type());
    const int64_t S = (int64_t)types.size();
    TORCH_CHECK(dasp_sos_supported_sections((int)S), "dasp::parametric_eq_norm: ", S, " sections are not supported (2, 4, 6, 8)");
    TORCH_CHECK(pn.dim() == 2 && pn.size(1) == 3 * S && (pn.size(0) == 1 || pn.size(0) == x.size(0)),
                "dasp::parametric_eq_norm: parameters must be (", x.size(0), " or 1, ", 3 * S, "), got ", pn.sizes());
    TORCH_CHECK((int64_t)lo.size() == 3 * S && (int64_t)span.size() == 3 * S, "dasp::parametric_eq_norm: lo / span need ", 3 * S, " entries");
    return PeqDims{x.size(0), x.size(1), x.size(2), pn.size(0), S};
}
// y, work32 = [tab | carries] (what the adjoint reads), work64 = [dtab | segtab]
std::tuple<Tensor, Tensor, Tensor> peq_norm_forward(const Tensor& x, const Tensor& pn, double sample_rate, at::IntArrayRef types,
                                                    at::ArrayRef<double> lo, at::ArrayRef<double> span, int64_t tseg, bool save,
                                                    const c10::optional<Tensor>& range_flag) {
    const PeqDims d = peq_check(x, pn, types, lo, span);
    c10::DeviceGuard guard(x.device());
    const Tensor x32 = x.contiguous(), pn32 = f32c(pn);
    Tensor y = at::empty_like(x32);
    const long n_tab = round64(d.Bp * dasp_sos_table_floats((int)d.S));
    const long n_car = save ? round64(dasp_sos_carry_floats(d.B * d.C, d.N, (int)d.S)) : 0;
    const long n_seg = tseg ? round64(dasp_sos_seg_floats(d.B * d.C, d.N, (int)d.S, tseg)) : 0;
    const long n_dt = d.Bp * dasp_sos_dtab_doubles((int)d.S), n_st = tseg ? d.Bp * dasp_sos_segtab_doubles((int)d.S) : 0;
    Tensor work32 = empty_f32(n_tab + n_car, x32);
    Tensor work64 = at::empty({n_dt + n_st}, x32.options().dtype(at::kDouble));
    if (x32.numel() == 0) return {y, work32, work64};
    Tensor segbuf = tseg ? empty_f32(n_seg, x32) : Tensor();              // scratch of the forward pre-pass only
    std::vector<int> ty(types.begin(), types.end());
    float* w = work32.data_ptr<float>();
    double* w64 = work64.data_ptr<double>();
    check_rc(dasp_peq_forward_norm(pn32.data_ptr<float>(), (int)d.Bp, (int)d.S, ty.data(), sample_rate, lo.data(), span.data(), flag_ptr(range_flag, x32), w, w64,
                                   x32.data_ptr<float>(), y.data_ptr<float>(), save ? w + n_tab : nullptr, (int)d.B, (int)d.C, d.N, tseg,
                                   tseg ? w64 + n_dt : nullptr, fp(segbuf), stream_of(x32)),
             "dasp_peq_forward_norm");
    return {y, work32, work64};
}
// -> (gx or empty, gp (Bp, 3 S) or empty). Reads the forward call's tables and chunk states; its own scratch (partial sums, segment
// pre-pass buffers) is allocated here, so the op leaves its inputs as it found them (the completion counter word inside `tab` is returned
// to zero by the kernels).
// mode 1: gp (Bp, 3 S), the layout of the normalised parameter tensor; mode 2: gp (3 S, Bp), one contiguous row per control tensor
std::tuple<Tensor, Tensor> peq_backward_impl(const Tensor& x, const Tensor& gy, const Tensor& work32, const Tensor& work64, int64_t Bp, int64_t S,
                                             int64_t tseg, bool need_gx, bool need_gp, int64_t mode) {
    need_device(x, "x");
    same_device(x, gy, "grad_output");
    TORCH_CHECK(gy.sizes() == x.sizes(), "dasp::_peq_norm_backward: grad_output ", gy.sizes(), " does not match x ", x.sizes());
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), C = x.size(1), N = x.size(2);
    const Tensor x32 = x.contiguous(), g32 = f32c(gy);
    Tensor gx = need_gx ? at::empty_like(x32) : at::empty({0}, x32.options());
    TORCH_CHECK(mode == 1 || mode == 2, "dasp::_peq_backward: gradient layout 1 (Bp, 3 S) or 2 (3 S, Bp)");
    Tensor gp = need_gp ? (mode == 1 ? at::empty({B, S, 3}, x32.options()) : at::empty({3 * S, B}, x32.options())) : at::empty({0}, x32.options());
    if (x32.numel() == 0 || (!need_gx && !need_gp))
        return {gx, need_gp ? (mode == 1 ? at::zeros({Bp, 3 * S}, x32.options()) : at::zeros({3 * S, Bp}, x32.options())) : gp};
    const long n_tab = round64(Bp * dasp_sos_table_floats((int)S));
    const long n_dt = Bp * dasp_sos_dtab_doubles((int)S);
    const long G = dasp_sos_segments(N, tseg);
    TORCH_CHECK(work32.numel() >= n_tab + round64(dasp_sos_carry_floats(B * C, N, (int)S)) && work64.numel() >= n_dt + (tseg ? Bp * dasp_sos_segtab_doubles((int)S) : 0),
                "dasp::_peq_norm_backward: work buffers do not belong to a forward call of this shape");
    Tensor partials = need_gp ? empty_f32(round64(dasp_sos_partial_floats(B * C * G, (int)S)), x32) : Tensor();
    Tensor segbuf = tseg ? empty_f32(round64(dasp_sos_seg_floats(B * C, N, (int)S, tseg)), x32) : Tensor();
    float* w = work32.data_ptr<float>();
    double* w64 = work64.data_ptr<double>();
    // mode 1: gout (B, S, 3) = the layout of the (Bp, 3 S) parameter tensor; mode 2: gout (3 S, B)
    check_rc(dasp_peq_backward(w, w64, (int)Bp, x32.data_ptr<float>(), g32.data_ptr<float>(), w + n_tab, need_gx ? gx.data_ptr<float>() : nullptr,
                               fp(partials), (int)mode, need_gp ? gp.data_ptr<float>() : nullptr, (int)B, (int)C, N, (int)S, tseg,
                               tseg ? w64 + n_dt : nullptr, fp(segbuf), stream_of(x32)),
             "dasp_peq_backward");
    if (need_gp) {
        if (Bp == 1 && B != 1) gp = gp.sum(mode == 1 ? 0 : 1, /*keepdim=*/true);   // one filter set shared by the batch (functional.py:208-220)
        if (mode == 1) gp = gp.reshape({Bp, 3 * S});
    }
    return {gx, gp};
}
std::tuple<Tensor, Tensor> peq_norm_backward(const Tensor& x, const Tensor& gy, const Tensor& work32, const Tensor& work64, int64_t Bp, int64_t S,
                                             int64_t tseg, bool need_gx, bool need_gp) {
    return peq_backward_impl(x, gy, work32, work64, Bp, S, tseg, need_gx, need_gp, 1);
}
Tensor peq_norm_device(const Tensor& x, const Tensor& pn, double sample_rate, at::IntArrayRef types, at::ArrayRef<double> lo, at::ArrayRef<double> span,
                       const c10::optional<Tensor>& range_flag) {
    const PeqDims d = peq_check(x, pn, types, lo, span);
    return std::get<0>(peq_norm_forward(x, pn, sample_rate, types, lo, span, sos_segment_tiles(d.B * d.C, d.N), false, range_flag));
}
struct PeqNormFn : public torch::autograd::Function<PeqNormFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& pn, double sample_rate, std::vector<int64_t> types,
                          std::vector<double> lo, std::vector<double> span, const c10::optional<Tensor>& range_flag) {
        const PeqDims d = peq_check(x, pn, types, lo, span);
        const int64_t tseg = sos_segment_tiles(d.B * d.C, d.N);
        const bool need = x.requires_grad() || pn.requires_grad();
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_peq_norm_forward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, double, at::IntArrayRef, at::ArrayRef<double>,
                                                                       at::ArrayRef<double>, int64_t, bool, const c10::optional<Tensor>&)>();
        auto [y, w32, w64] = op.call(x, pn, sample_rate, types, lo, span, tseg, need, range_flag);
        if (need) {
            ctx->save_for_backward({x, w32, w64});
            ctx->saved_data["Bp"] = d.Bp; ctx->saved_data["S"] = d.S; ctx->saved_data["tseg"] = tseg;
            ctx->saved_data["pn_dtype"] = (int64_t)pn.scalar_type();
        }
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_peq_norm_backward", "")
                             .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, int64_t, bool, bool)>();
        const bool need_gx = ctx->needs_input_grad(0), need_gp = ctx->needs_input_grad(1);
        auto [gx, gp] = op.call(saved[0], grads[0], saved[1], saved[2], ctx->saved_data["Bp"].toInt(), ctx->saved_data["S"].toInt(),
                                ctx->saved_data["tseg"].toInt(), need_gx, need_gp);
        if (need_gp) gp = gp.to((at::ScalarType)ctx->saved_data["pn_dtype"].toInt());
        return {need_gx ? gx : Tensor(), need_gp ? gp : Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};
Tensor peq_norm_autograd(const Tensor& x, const Tensor& pn, double sample_rate, at::IntArrayRef types, at::ArrayRef<double> lo, at::ArrayRef<double> span,
                         const c10::optional<Tensor>& range_flag) {
    return PeqNormFn::apply(x, pn, sample_rate, types.vec(), lo.vec(), span.vec(), range_flag);
}

// ---- compressor / expander on (bs, 5) control rows (ops.DynamicsCtlFunction; dasp_dynamics_forward(_seg) / _backward(_seg)) -------------
int64_t dyn_segment_tiles(int64_t B, int64_t N) {
    return g_dyn_tiles_override >= 0 ? g_dyn_tiles_override : dasp_dyn_segment_tiles(B, N);
}
void dyn_check(const Tensor& x, const Tensor& ctl, int64_t mode, int64_t lookahead) {
    need_device(x, "x");
    same_device(x, ctl, "ctl");
    TORCH_CHECK(x.dim() == 3 && x.scalar_type() == at::kFloat, "dasp::dynamics_ctl: x must be float32 (bs, chs, seq_len), got ", x.scalar_type(), " ", x.sizes());
    // the kernels read five controls per batch item at ctl[b * 5 ...] (the reference does not broadcast a parameter batch of 1 either,
    // functional.py:330-336)
    TORCH_CHECK(ctl.dim() == 2 && ctl.size(0) == x.size(0) && ctl.size(1) == 5, "The size of tensor a (", ctl.dim() ? ctl.size(0) : 1,
                ") must match the size of tensor b (", x.size(0), ") at non-singleton dimension 0 (ctl must be (", x.size(0), ", 5), got ", ctl.sizes(), ")");
    TORCH_CHECK(mode == 0 || mode == 1, "dasp::dynamics_ctl: mode 0 (compressor) or 1 (expander)");
    TORCH_CHECK(lookahead >= 0, "dasp::dynamics_ctl: lookahead_samples must be >= 0");
}
// -> y, carries, lin (the linear gain curve, kept only with a look-ahead delay)
std::tuple<Tensor, Tensor, Tensor> dyn_forward(const Tensor& x, const Tensor& ctl, int64_t mode, double sample_rate, double eps, int64_t lookahead,
                                               int64_t tseg, bool save) {
    dyn_check(x, ctl, mode, lookahead);
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), C = x.size(1), N = x.size(2);
    const Tensor x32 = x.contiguous(), c32 = f32c(ctl);
    Tensor y = at::empty_like(x32);
    Tensor carries = empty_f32(save && x32.numel() ? dasp_dyn_carry_floats(B, N) : 0, x32);
    Tensor lin = lookahead > 0 ? at::empty({B, N}, x32.options()) : at::empty({0}, x32.options());
    if (x32.numel() == 0) return {y, carries, lin};
    if (tseg) {
        Tensor segbuf = empty_f32(2 * B * dasp_dyn_segments(N, tseg), x32);
        Tensor counters = dyn_counters(x32, B);
        check_rc(dasp_dynamics_forward_seg((int)mode, x32.data_ptr<float>(), c32.data_ptr<float>(), y.data_ptr<float>(), fp(carries), fp(lin), fp(segbuf),
                                           (int)B, (int)C, N, sample_rate, (float)eps, (int)lookahead, tseg, counters.data_ptr<int>(), stream_of(x32)),
                 "dasp_dynamics_forward_seg");
    } else {
        check_rc(dasp_dynamics_forward((int)mode, x32.data_ptr<float>(), c32.data_ptr<float>(), y.data_ptr<float>(), fp(carries), fp(lin), (int)B, (int)C, N,
                                       sample_rate, (float)eps, (int)lookahead, stream_of(x32)),
                 "dasp_dynamics_forward");
    }
    return {y, carries, lin};
}
std::tuple<Tensor, Tensor> dyn_backward(const Tensor& x, const Tensor& ctl, const Tensor& gy, const Tensor& carries, const Tensor& lin, int64_t mode,
                                        double sample_rate, double eps, int64_t lookahead, int64_t tseg) {
    dyn_check(x, ctl, mode, lookahead);
    same_device(x, gy, "grad_output");
    TORCH_CHECK(gy.sizes() == x.sizes(), "dasp::_dynamics_backward: grad_output ", gy.sizes(), " does not match x ", x.sizes());
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), C = x.size(1), N = x.size(2);
    const Tensor x32 = x.contiguous(), c32 = f32c(ctl), g32 = f32c(gy);
    Tensor gx = at::empty_like(x32), gctl = at::empty({B, 5}, x32.options());
    if (x32.numel() == 0) return {gx, gctl.zero_()};
    TORCH_CHECK(carries.numel() >= dasp_dyn_carry_floats(B, N), "dasp::_dynamics_backward: `carries` does not belong to a forward call of this shape");
    const long G = dasp_dyn_segments(N, tseg);
    Tensor partials = empty_f32(dasp_dyn_partial_floats(B * G), x32);
    if (tseg) {
        Tensor segbuf = empty_f32(2 * B * G, x32);
        Tensor counters = dyn_counters(x32, B);
        check_rc(dasp_dynamics_backward_seg((int)mode, x32.data_ptr<float>(), c32.data_ptr<float>(), g32.data_ptr<float>(), fp(carries), lookahead > 0 ? fp(lin) : nullptr,
                                            gx.data_ptr<float>(), gctl.data_ptr<float>(), fp(partials), fp(segbuf), (int)B, (int)C, N, sample_rate, (float)eps,
                                            (int)lookahead, tseg, counters.data_ptr<int>(), stream_of(x32)),
                 "dasp_dynamics_backward_seg");
    } else {
        check_rc(dasp_dynamics_backward((int)mode, x32.data_ptr<float>(), c32.data_ptr<float>(), g32.data_ptr<float>(), fp(carries), lookahead > 0 ? fp(lin) : nullptr,
                                        gx.data_ptr<float>(), gctl.data_ptr<float>(), fp(partials), (int)B, (int)C, N, sample_rate, (float)eps, (int)lookahead,
                                        stream_of(x32)),
                 "dasp_dynamics_backward");
    }
    return {gx, gctl};
}
// The same on the reference's own six control tensors (functional.py:275-286), straight from where they lie: five device vectors in,
// the six gradients as the rows of one (6, bs) tensor out (release_ms: zeros, written by the kernel) - no stacking launch in front of the
// kernels, no transposition or fill behind them (dasp_dynamics_forward_rows / _backward_rows).
void dyn6_check(const Tensor& x, at::TensorList five, int64_t mode, int64_t lookahead) {
    need_device(x, "x");
    TORCH_CHECK(x.dim() == 3 && x.scalar_type() == at::kFloat, "dasp::dynamics: x must be float32 (bs, chs, seq_len), got ", x.scalar_type(), " ", x.sizes());
    TORCH_CHECK(five.size() == 5, "dasp::_dynamics6: five control vectors (threshold_db, ratio, attack_ms, knee_db, makeup_gain_db), got ", five.size());
    for (const Tensor& c : five) {
        same_device(x, c, "control");
        TORCH_CHECK(c.dim() == 1 && c.scalar_type() == at::kFloat && c.is_contiguous() && c.sym_numel().guard_int(__FILE__, __LINE__) == x.size(0),
                    "dasp::_dynamics6: every control is a contiguous float32 vector of bs = ", x.size(0), " values, got ", c.scalar_type(), " ", c.sizes());
    }
    TORCH_CHECK(mode == 0 || mode == 1, "dasp::dynamics: mode 0 (compressor) or 1 (expander)");
    TORCH_CHECK(lookahead >= 0, "dasp::dynamics: lookahead_samples must be >= 0");
}
std::tuple<Tensor, Tensor, Tensor> dyn6_forward(const Tensor& x, at::TensorList five, int64_t mode, double sample_rate, double eps, int64_t lookahead,
                                                int64_t tseg, bool save) {
    dyn6_check(x, five, mode, lookahead);
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), C = x.size(1), N = x.size(2);
    const Tensor x32 = x.contiguous();
    Tensor y = at::empty_like(x32);
    Tensor carries = empty_f32(save && x32.numel() ? dasp_dyn_carry_floats(B, N) : 0, x32);
    Tensor lin = lookahead > 0 ? at::empty({B, N}, x32.options()) : at::empty({0}, x32.options());
    if (x32.numel() == 0) return {y, carries, lin};
    const float* rows[5];
    for (int i = 0; i < 5; ++i) rows[i] = five[i].data_ptr<float>();
    Tensor segbuf = tseg ? empty_f32(2 * B * dasp_dyn_segments(N, tseg), x32) : Tensor();
    Tensor counters = tseg ? dyn_counters(x32, B) : Tensor();
    check_rc(dasp_dynamics_forward_rows((int)mode, x32.data_ptr<float>(), rows, y.data_ptr<float>(), fp(carries), fp(lin), tseg ? fp(segbuf) : nullptr, (int)B, (int)C, N,
                                        sample_rate, (float)eps, (int)lookahead, tseg, tseg ? counters.data_ptr<int>() : nullptr, stream_of(x32)),
             "dasp_dynamics_forward_rows");
    return {y, carries, lin};
}
std::tuple<Tensor, Tensor> dyn6_backward(const Tensor& x, at::TensorList five, const Tensor& gy, const Tensor& carries, const Tensor& lin, int64_t mode,
                                         double sample_rate, double eps, int64_t lookahead, int64_t tseg) {
    dyn6_check(x, five, mode, lookahead);
    same_device(x, gy, "grad_output");
    TORCH_CHECK(gy.sizes() == x.sizes(), "dasp::_dynamics6_backward: grad_output ", gy.sizes(), " does not match x ", x.sizes());
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), C = x.size(1), N = x.size(2);
    const Tensor x32 = x.contiguous(), g32 = f32c(gy);
    Tensor gx = at::empty_like(x32), g6 = at::empty({6, B}, x32.options());
    if (x32.numel() == 0) return {gx, g6.zero_()};
    TORCH_CHECK(carries.numel() >= dasp_dyn_carry_floats(B, N), "dasp::_dynamics6_backward: `carries` does not belong to a forward call of this shape");
    const long G = dasp_dyn_segments(N, tseg);
    Tensor partials = empty_f32(dasp_dyn_partial_floats(B * G), x32);
    const float* rows[5];
    for (int i = 0; i < 5; ++i) rows[i] = five[i].data_ptr<float>();
    float* grows[6];
    for (int i = 0; i < 6; ++i) grows[i] = g6.data_ptr<float>() + i * B;
    Tensor segbuf = tseg ? empty_f32(2 * B * G, x32) : Tensor();
    Tensor counters = tseg ? dyn_counters(x32, B) : Tensor();
    check_rc(dasp_dynamics_backward_rows((int)mode, x32.data_ptr<float>(), rows, g32.data_ptr<float>(), fp(carries), lookahead > 0 ? fp(lin) : nullptr, gx.data_ptr<float>(),
                                         grows, fp(partials), tseg ? fp(segbuf) : nullptr, (int)B, (int)C, N, sample_rate, (float)eps, (int)lookahead, tseg,
                                         tseg ? counters.data_ptr<int>() : nullptr, stream_of(x32)),
             "dasp_dynamics_backward_rows");
    return {gx, g6};
}
Tensor dyn_device(const Tensor& x, const Tensor& ctl, int64_t mode, double sample_rate, double eps, int64_t lookahead) {
    dyn_check(x, ctl, mode, lookahead);
    return std::get<0>(dyn_forward(x, ctl, mode, sample_rate, eps, lookahead, dyn_segment_tiles(x.size(0), x.size(2)), false));
}
struct DynFn : public torch::autograd::Function<DynFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& ctl, int64_t mode, double sample_rate, double eps, int64_t lookahead) {
        dyn_check(x, ctl, mode, lookahead);
        const int64_t tseg = dyn_segment_tiles(x.size(0), x.size(2));
        const bool need = x.requires_grad() || ctl.requires_grad();
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_dynamics_forward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, int64_t, double, double, int64_t, int64_t, bool)>();
        auto [y, carries, lin] = op.call(x, ctl, mode, sample_rate, eps, lookahead, tseg, need);
        if (need) {
            ctx->save_for_backward({x, ctl, carries, lin});
            ctx->saved_data["mode"] = mode; ctx->saved_data["sr"] = sample_rate; ctx->saved_data["eps"] = eps;
            ctx->saved_data["look"] = lookahead; ctx->saved_data["tseg"] = tseg; ctx->saved_data["ctl_dtype"] = (int64_t)ctl.scalar_type();
        }
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto s = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_dynamics_backward", "")
                             .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, double, double,
                                                               int64_t, int64_t)>();
        auto [gx, gctl] = op.call(s[0], s[1], grads[0], s[2], s[3], ctx->saved_data["mode"].toInt(), ctx->saved_data["sr"].toDouble(),
                                  ctx->saved_data["eps"].toDouble(), ctx->saved_data["look"].toInt(), ctx->saved_data["tseg"].toInt());
        return {ctx->needs_input_grad(0) ? gx : Tensor(),
                ctx->needs_input_grad(1) ? gctl.to((at::ScalarType)ctx->saved_data["ctl_dtype"].toInt()) : Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};
Tensor dyn_autograd(const Tensor& x, const Tensor& ctl, int64_t mode, double sample_rate, double eps, int64_t lookahead) {
    return DynFn::apply(x, ctl, mode, sample_rate, eps, lookahead);
}

// ---- EQ -> compressor as ONE forward pass that saves for both backward passes (csrc/chainfwd.hip dasp_chain_forward_saving; SURVEY 8(f2) on
// the pass that carries gradients, examples/style_transfer.py:150-154). Forward: the EQ's design launch + one pass over x writing y, the
// EQ's output (the compressor's input), the EQ's chunk states and the compressor's tile carries. Backward: the two existing backward
// passes, the compressor's on the saved EQ output, the EQ's on what that returns. One workgroup per item: the callers take it from 384
// rows on (profiles/r06/chain_fwd_saving_ab.log, chain_step_ab.log), below that the two forward launches are as fast or faster.
// -> y, yeq, work32 = [tab | eq carries], work64 = [dtab], dyn carries
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> eq_dyn_norm_forward(const Tensor& x, const Tensor& pn, double sample_rate, at::IntArrayRef types,
                                                                       at::ArrayRef<double> lo, at::ArrayRef<double> span, const Tensor& ctl, int64_t mode,
                                                                       double eps, const c10::optional<Tensor>& range_flag) {
    const PeqDims d = peq_check(x, pn, types, lo, span);
    dyn_check(x, ctl, mode, 0);
    TORCH_CHECK(d.S == 6 && d.C <= 2, "dasp::eq_dyn_norm: six sections and at most two channels (got ", d.S, ", ", d.C, ")");
    c10::DeviceGuard guard(x.device());
    const Tensor x32 = x.contiguous(), pn32 = f32c(pn), c32 = f32c(ctl);
    Tensor y = at::empty_like(x32), yeq = at::empty_like(x32);
    const long n_tab = round64(d.Bp * dasp_sos_table_floats((int)d.S));
    const long n_car = round64(dasp_sos_carry_floats(d.B * d.C, d.N, (int)d.S));
    Tensor work32 = empty_f32(n_tab + n_car, x32);
    Tensor work64 = at::empty({d.Bp * dasp_sos_dtab_doubles((int)d.S)}, x32.options().dtype(at::kDouble));
    Tensor dcar = empty_f32(x32.numel() ? dasp_dyn_carry_floats(d.B, d.N) : 0, x32);
    if (x32.numel() == 0) return {y, yeq, work32, work64, dcar};
    std::vector<int> ty(types.begin(), types.end());
    float* w = work32.data_ptr<float>();
    check_rc(dasp_peq_prepare_norm(pn32.data_ptr<float>(), (int)d.Bp, (int)d.S, ty.data(), sample_rate, lo.data(), span.data(), flag_ptr(range_flag, x32), w,
                                   work64.data_ptr<double>(), stream_of(x32)),
             "dasp_peq_prepare_norm");
    check_rc(dasp_chain_forward_saving(w, (int)d.Bp, x32.data_ptr<float>(), c32.data_ptr<float>(), y.data_ptr<float>(), yeq.data_ptr<float>(), w + n_tab,
                                       dcar.data_ptr<float>(), (int)d.B, (int)d.C, d.N, (int)d.S, (int)mode, sample_rate, (float)eps, stream_of(x32)),
             "dasp_chain_forward_saving");
    return {y, yeq, work32, work64, dcar};
}
Tensor eq_dyn_norm_device(const Tensor& x, const Tensor& pn, double sample_rate, at::IntArrayRef types, at::ArrayRef<double> lo, at::ArrayRef<double> span,
                          const Tensor& ctl, int64_t mode, double eps, const c10::optional<Tensor>& range_flag) {
    return std::get<0>(eq_dyn_norm_forward(x, pn, sample_rate, types, lo, span, ctl, mode, eps, range_flag));
}
struct EqDynNormFn : public torch::autograd::Function<EqDynNormFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& pn, double sample_rate, std::vector<int64_t> types, std::vector<double> lo,
                          std::vector<double> span, const Tensor& ctl, int64_t mode, double eps, const c10::optional<Tensor>& range_flag) {
        const PeqDims d = peq_check(x, pn, types, lo, span);
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_eq_dyn_norm_forward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, double, at::IntArrayRef, at::ArrayRef<double>,
                                                                                       at::ArrayRef<double>, const Tensor&, int64_t, double,
                                                                                       const c10::optional<Tensor>&)>();
        auto [y, yeq, w32, w64, dcar] = op.call(x, pn, sample_rate, types, lo, span, ctl, mode, eps, range_flag);
        if (x.requires_grad() || pn.requires_grad() || ctl.requires_grad()) {
            ctx->save_for_backward({x, yeq, w32, w64, dcar, ctl});
            ctx->saved_data["Bp"] = d.Bp; ctx->saved_data["S"] = d.S; ctx->saved_data["mode"] = mode; ctx->saved_data["sr"] = sample_rate;
            ctx->saved_data["eps"] = eps; ctx->saved_data["pn_dtype"] = (int64_t)pn.scalar_type(); ctx->saved_data["ctl_dtype"] = (int64_t)ctl.scalar_type();
        }
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto s = ctx->get_saved_variables();
        static auto dyn_op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_dynamics_backward", "")
                                 .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, double, double,
                                                                   int64_t, int64_t)>();
        static auto peq_op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_peq_norm_backward", "")
                                 .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, int64_t, bool, bool)>();
        const bool need_gx = ctx->needs_input_grad(0), need_gp = ctx->needs_input_grad(1), need_gc = ctx->needs_input_grad(2);       // (edges are counted over the tensor arguments: x, param_tensor, ctl)
        // compressor backward on the saved EQ output (its input), one workgroup per item as the forward pass ran
        auto [geq, gctl] = dyn_op.call(s[1], s[5], grads[0], s[4], at::empty({0}, s[1].options()), ctx->saved_data["mode"].toInt(), ctx->saved_data["sr"].toDouble(),
                                       ctx->saved_data["eps"].toDouble(), 0, 0);
        Tensor gx, gp;
        if (need_gx || need_gp) {
            auto r = peq_op.call(s[0], geq, s[2], s[3], ctx->saved_data["Bp"].toInt(), ctx->saved_data["S"].toInt(), 0, need_gx, need_gp);
            gx = std::get<0>(r); gp = std::get<1>(r);
            if (need_gp) gp = gp.to((at::ScalarType)ctx->saved_data["pn_dtype"].toInt());
        }
        return {need_gx ? gx : Tensor(), need_gp ? gp : Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                need_gc ? gctl.to((at::ScalarType)ctx->saved_data["ctl_dtype"].toInt()) : Tensor(), Tensor(), Tensor(), Tensor()};
    }
};
Tensor eq_dyn_norm_autograd(const Tensor& x, const Tensor& pn, double sample_rate, at::IntArrayRef types, at::ArrayRef<double> lo, at::ArrayRef<double> span,
                            const Tensor& ctl, int64_t mode, double eps, const c10::optional<Tensor>& range_flag) {
    return EqDynNormFn::apply(x, pn, sample_rate, types.vec(), lo.vec(), span.vec(), ctl, mode, eps, range_flag);
}

// ---- the chain's control de-normalisation (ops.ChainControlsFunction; dasp_chain_controls / _backward) -------------------------------------
// lo, span: 32 floats each (compressor 0-5, reverb 6-30, gain 31)
std::tuple<Tensor, Tensor, Tensor, Tensor> chain_controls(const Tensor& comp_pn, const Tensor& reverb_pn, const Tensor& gain_pn, at::ArrayRef<double> lo,
                                                          at::ArrayRef<double> span, const c10::optional<Tensor>& range_flag) {
    need_device(comp_pn, "comp_params");
    same_device(comp_pn, reverb_pn, "reverb_params");
    same_device(comp_pn, gain_pn, "gain_params");
    const int64_t B = comp_pn.size(0);
    TORCH_CHECK(comp_pn.dim() == 2 && comp_pn.size(1) == 6 && reverb_pn.dim() == 2 && reverb_pn.size(0) == B && reverb_pn.size(1) == 25 && gain_pn.dim() == 2 &&
                    gain_pn.size(0) == B && gain_pn.size(1) == 1,
                "dasp::chain_controls: parameters must be (bs, 6), (bs, 25), (bs, 1), got ", comp_pn.sizes(), " ", reverb_pn.sizes(), " ", gain_pn.sizes());
    TORCH_CHECK(lo.size() == 32 && span.size() == 32, "dasp::chain_controls: lo / span need 32 entries");
    c10::DeviceGuard guard(comp_pn.device());
    const auto o = comp_pn.options().dtype(at::kFloat);
    Tensor ctl = at::empty({B, 5}, o), gains = at::empty({B, 12}, o), decays = at::empty({B, 12}, o), mix = at::empty({B}, o);
    if (B) {
        float lof[32], spf[32];
        for (int i = 0; i < 32; ++i) { lof[i] = (float)lo[i]; spf[i] = (float)span[i]; }
        const Tensor c = f32c(comp_pn), r = f32c(reverb_pn), g = f32c(gain_pn);
        check_rc(dasp_chain_controls(c.data_ptr<float>(), r.data_ptr<float>(), g.data_ptr<float>(), lof, spf, ctl.data_ptr<float>(), gains.data_ptr<float>(),
                                     decays.data_ptr<float>(), mix.data_ptr<float>(), flag_ptr(range_flag, comp_pn), (int)B, stream_of(comp_pn)),
                 "dasp_chain_controls");
    }
    return {ctl, gains, decays, mix};
}
std::tuple<Tensor, Tensor, Tensor> chain_controls_backward(const Tensor& gctl, const Tensor& ggain, const Tensor& gdecay, const Tensor& gmix, at::ArrayRef<double> span) {
    need_device(gctl, "grad ctl");
    const int64_t B = gctl.size(0);
    TORCH_CHECK(gctl.dim() == 2 && gctl.size(1) == 5 && ggain.sizes() == at::IntArrayRef({B, 12}) && gdecay.sizes() == at::IntArrayRef({B, 12}) && gmix.numel() == B,
                "dasp::_chain_controls_backward: gradients must be (bs, 5), (bs, 12), (bs, 12), (bs)");
    TORCH_CHECK(span.size() == 32, "dasp::_chain_controls_backward: span needs 32 entries");
    c10::DeviceGuard guard(gctl.device());
    const auto o = gctl.options().dtype(at::kFloat);
    Tensor gc = at::empty({B, 6}, o), gr = at::empty({B, 25}, o), gg = at::empty({B, 1}, o);
    if (B) {
        float spf[32];
        for (int i = 0; i < 32; ++i) spf[i] = (float)span[i];
        const Tensor a = f32c(gctl), b = f32c(ggain), c = f32c(gdecay), d = f32c(gmix);
        check_rc(dasp_chain_controls_backward(a.data_ptr<float>(), b.data_ptr<float>(), c.data_ptr<float>(), d.data_ptr<float>(), spf, gc.data_ptr<float>(),
                                              gr.data_ptr<float>(), gg.data_ptr<float>(), (int)B, stream_of(gctl)),
                 "dasp_chain_controls_backward");
    }
    return {gc, gr, gg};
}
struct ChainControlsFn : public torch::autograd::Function<ChainControlsFn> {
    static variable_list forward(AutogradContext* ctx, const Tensor& comp_pn, const Tensor& reverb_pn, const Tensor& gain_pn, std::vector<double> lo,
                                 std::vector<double> span, const c10::optional<Tensor>& range_flag) {
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::chain_controls", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, at::ArrayRef<double>, at::ArrayRef<double>,
                                                                               const c10::optional<Tensor>&)>();
        auto [ctl, gains, decays, mix] = op.call(comp_pn, reverb_pn, gain_pn, lo, span, range_flag);
        ctx->saved_data["span"] = span;
        ctx->saved_data["B"] = comp_pn.size(0);
        ctx->saved_data["dt"] = std::vector<int64_t>{(int64_t)comp_pn.scalar_type(), (int64_t)reverb_pn.scalar_type(), (int64_t)gain_pn.scalar_type()};
        return {ctl, gains, decays, mix};
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        const int64_t B = ctx->saved_data["B"].toInt();
        const auto span = ctx->saved_data["span"].toDoubleVector();
        const auto dt = ctx->saved_data["dt"].toIntVector();
        Tensor any;
        for (const auto& t : g) if (t.defined()) { any = t; break; }
        TORCH_CHECK(any.defined(), "dasp::chain_controls backward without any gradient");
        const auto o = any.options().dtype(at::kFloat);
        auto z = [&](const Tensor& t, at::IntArrayRef shape) { return t.defined() ? t : at::zeros(shape, o); };
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_chain_controls_backward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, at::ArrayRef<double>)>();
        auto [gc, gr, gg] = op.call(z(g[0], {B, 5}), z(g[1], {B, 12}), z(g[2], {B, 12}), z(g[3], {B}), span);
        return {gc.to((at::ScalarType)dt[0]), gr.to((at::ScalarType)dt[1]), gg.to((at::ScalarType)dt[2]), Tensor(), Tensor(), Tensor()};
    }
};
std::tuple<Tensor, Tensor, Tensor, Tensor> chain_controls_autograd(const Tensor& comp_pn, const Tensor& reverb_pn, const Tensor& gain_pn, at::ArrayRef<double> lo,
                                                                   at::ArrayRef<double> span, const c10::optional<Tensor>& range_flag) {
    auto r = ChainControlsFn::apply(comp_pn, reverb_pn, gain_pn, lo.vec(), span.vec(), range_flag);
    return {r[0], r[1], r[2], r[3]};
}

// ---- noise-shaped reverb on control matrices (ops.ReverbFunction; dasp_reverb_forward(_rng) / _backward(_rng)) ---------------------------
// noise: (2 bs, nb, L + taps - 1) or undefined = generated inside the filter-bank kernels from `seed` (+ the device word seed_offset);
// fspec: dasp_reverb_filter_spectrum of the (nb, taps) bank (cached by the Python side); gains / decays (bs, nb), mix (bs).
struct RvDims { int64_t B, C, N; long sizes[14]; };
RvDims reverb_check(const Tensor& x, const c10::optional<Tensor>& noise, const Tensor& fspec, const Tensor& gains, const Tensor& decays, const Tensor& mix, int64_t L,
                    int64_t taps, int64_t nb, const c10::optional<Tensor>& seed_offset) {
    need_device(x, "x");
    same_device(x, fspec, "filter spectra");
    same_device(x, gains, "band gains"); same_device(x, decays, "band decays"); same_device(x, mix, "mix");
    TORCH_CHECK(x.dim() == 3 && x.scalar_type() == at::kFloat, "dasp::reverb: x must be float32 (bs, chs, seq_len), got ", x.scalar_type(), " ", x.sizes());
    TORCH_CHECK(x.size(1) == 1 || x.size(1) == 2, "noise_shaped_reverberation takes mono or stereo input, got ", x.size(1), " channels");
    RvDims d{x.size(0), x.size(1), x.size(2), {}};
    // the kernels index gains[b * nb + band]: a (k, nb) stack with k != bs must not be reshaped silently (functional.py:498-544)
    TORCH_CHECK(gains.dim() == 2 && gains.size(0) == d.B && gains.size(1) == nb && decays.sizes() == gains.sizes() && mix.numel() == d.B, "shape '[", d.B, ", ", nb,
                "]' is invalid for band gains / decays of shapes ", gains.sizes(), " / ", decays.sizes(), " and mix with ", mix.numel(), " values");
    if (x.numel()) check_rc(dasp_reverb_sizes((int)d.B, d.N, (int)L, (int)taps, (int)nb, d.sizes), "dasp_reverb_sizes");
    if (x.numel()) TORCH_CHECK(fspec.scalar_type() == at::kFloat && fspec.numel() >= 2 * d.sizes[4], "dasp::reverb: `fspec` is not the filter spectrum of this (nb, taps)");
    if (noise.has_value() && noise->defined()) {
        same_device(x, *noise, "noise");
        TORCH_CHECK(noise->numel() == 2 * d.B * nb * (L + taps - 1), "noise must hold (2 * ", d.B, ", ", nb, ", ", L + taps - 1, ") values, got ", noise->sizes());
        TORCH_CHECK(!(seed_offset.has_value() && seed_offset->defined()), "noise_seed_offset only applies to the generated noise");
    }
    if (seed_offset.has_value() && seed_offset->defined())
        TORCH_CHECK(seed_offset->is_cuda() && seed_offset->scalar_type() == at::kLong && seed_offset->numel() == 1 && seed_offset->device() == x.device(),
                    "noise_seed_offset must be a 1-element int64 tensor on x's device");
    return d;
}
Tensor cbuf(long n_complex, const Tensor& like) { return empty_f32(2 * n_complex, like); }
const unsigned long long* seed_ptr(const c10::optional<Tensor>& t) {
    return t.has_value() && t->defined() ? reinterpret_cast<const unsigned long long*>(t->data_ptr<int64_t>()) : nullptr;
}
// -> y (bs, 2, N), A, H, ir  (A: column transforms of x, H: the impulse responses' spectra, ir: the impulse responses; kept for the adjoint)
std::tuple<Tensor, Tensor, Tensor, Tensor> reverb_forward(const Tensor& x, const c10::optional<Tensor>& noise, const Tensor& fspec, const Tensor& gains, const Tensor& decays,
                                                          const Tensor& mix, int64_t L, int64_t taps, int64_t nb, int64_t seed, const c10::optional<Tensor>& seed_offset,
                                                          double decay_bound, bool save) {
    const RvDims d = reverb_check(x, noise, fspec, gains, decays, mix, L, taps, nb, seed_offset);
    c10::DeviceGuard guard(x.device());
    const Tensor x32 = x.contiguous();
    Tensor y = at::empty({d.B, 2, d.N}, x32.options());
    if (x32.numel() == 0) return {y, cbuf(0, x32), cbuf(0, x32), empty_f32(0, x32)};
    const Tensor g32 = f32c(gains), d32 = f32c(decays), m32 = f32c(mix.reshape({d.B}));
    Tensor A = cbuf(save ? d.sizes[6] : 0, x32), W2 = cbuf(save ? 0 : d.sizes[12], x32);
    Tensor W = cbuf(d.sizes[12], x32), H = cbuf(d.sizes[7], x32), Ah = cbuf(d.sizes[13], x32), ir = empty_f32(d.sizes[8], x32);
    if (noise.has_value() && noise->defined()) {
        const Tensor n32 = f32c(*noise);
        check_rc(dasp_reverb_forward(x32.data_ptr<float>(), n32.data_ptr<float>(), fspec.data_ptr<float>(), g32.data_ptr<float>(), d32.data_ptr<float>(),
                                     m32.data_ptr<float>(), y.data_ptr<float>(), fp(A), fp(H), fp(W), fp(W2), fp(Ah), fp(ir), (int)d.B, (int)d.C, d.N, (int)L, (int)taps,
                                     (int)nb, (float)decay_bound, stream_of(x32)),
                 "dasp_reverb_forward");
    } else {
        check_rc(dasp_reverb_forward_rng(x32.data_ptr<float>(), (unsigned long long)seed, seed_ptr(seed_offset), fspec.data_ptr<float>(), g32.data_ptr<float>(),
                                         d32.data_ptr<float>(), m32.data_ptr<float>(), y.data_ptr<float>(), fp(A), fp(H), fp(W), fp(W2), fp(Ah), fp(ir), (int)d.B, (int)d.C,
                                         d.N, (int)L, (int)taps, (int)nb, (float)decay_bound, stream_of(x32)),
                 "dasp_reverb_forward_rng");
    }
    return {y, A, H, ir};
}
// -> gx (bs, Cx, N), ggain (bs, nb), gdecay (bs, nb), gmix (bs)
std::tuple<Tensor, Tensor, Tensor, Tensor> reverb_backward(const Tensor& gy, const Tensor& ir, const Tensor& A, const Tensor& H, const c10::optional<Tensor>& noise,
                                                           const Tensor& fspec, const Tensor& gains, const Tensor& decays, const Tensor& mix, int64_t Cx, int64_t L,
                                                           int64_t taps, int64_t nb, int64_t seed, const c10::optional<Tensor>& seed_offset, double decay_bound) {
    need_device(gy, "grad_output");
    TORCH_CHECK(gy.dim() == 3 && gy.size(1) == 2, "dasp::_reverb_backward: grad_output must be (bs, 2, seq_len), got ", gy.sizes());
    const int64_t B = gy.size(0), N = gy.size(2);
    c10::DeviceGuard guard(gy.device());
    const auto o = gy.options().dtype(at::kFloat);
    Tensor ggain = at::empty({B, nb}, o), gdecay = at::empty({B, nb}, o), gmix = at::empty({B}, o);
    if (gy.numel() == 0) return {at::empty({B, Cx, N}, o), ggain.zero_(), gdecay.zero_(), gmix.zero_()};
    long sizes[14];
    check_rc(dasp_reverb_sizes((int)B, N, (int)L, (int)taps, (int)nb, sizes), "dasp_reverb_sizes");
    TORCH_CHECK(A.numel() >= 2 * sizes[6] && H.numel() >= 2 * sizes[7] && ir.numel() >= sizes[8], "dasp::_reverb_backward: A / H / ir do not belong to a forward call of this shape");
    const Tensor g32 = f32c(gy), ga = f32c(gains), de = f32c(decays), mi = f32c(mix.reshape({B}));
    Tensor gx = at::empty({B, 2, N}, o);
    Tensor Ag = cbuf(sizes[12], g32), W = cbuf(sizes[12], g32), P = cbuf(sizes[13], g32);
    Tensor gir = empty_f32(sizes[8], g32), part = empty_f32(sizes[11], g32), mix_part = empty_f32(sizes[10], g32);
    if (noise.has_value() && noise->defined()) {
        const Tensor n32 = f32c(*noise);
        check_rc(dasp_reverb_backward(ir.data_ptr<float>(), g32.data_ptr<float>(), n32.data_ptr<float>(), fspec.data_ptr<float>(), ga.data_ptr<float>(), de.data_ptr<float>(),
                                      mi.data_ptr<float>(), A.data_ptr<float>(), H.data_ptr<float>(), gx.data_ptr<float>(), ggain.data_ptr<float>(), gdecay.data_ptr<float>(),
                                      gmix.data_ptr<float>(), fp(Ag), fp(W), fp(P), fp(gir), fp(part), fp(mix_part), (int)B, (int)Cx, N, (int)L, (int)taps, (int)nb,
                                      (float)decay_bound, stream_of(g32)),
                 "dasp_reverb_backward");
    } else {
        check_rc(dasp_reverb_backward_rng(ir.data_ptr<float>(), g32.data_ptr<float>(), (unsigned long long)seed, seed_ptr(seed_offset), fspec.data_ptr<float>(),
                                          ga.data_ptr<float>(), de.data_ptr<float>(), mi.data_ptr<float>(), A.data_ptr<float>(), H.data_ptr<float>(), gx.data_ptr<float>(),
                                          ggain.data_ptr<float>(), gdecay.data_ptr<float>(), gmix.data_ptr<float>(), fp(Ag), fp(W), fp(P), fp(gir), fp(part), fp(mix_part),
                                          (int)B, (int)Cx, N, (int)L, (int)taps, (int)nb, (float)decay_bound, stream_of(g32)),
                 "dasp_reverb_backward_rng");
    }
    if (Cx == 1) gx = gx.sum(1, /*keepdim=*/true);             // the adjoint of the mono -> stereo duplication (functional.py:493-495)
    return {gx, ggain, gdecay, gmix};
}
Tensor reverb_device(const Tensor& x, const c10::optional<Tensor>& noise, const Tensor& fspec, const Tensor& gains, const Tensor& decays, const Tensor& mix, int64_t L,
                     int64_t taps, int64_t nb, int64_t seed, const c10::optional<Tensor>& seed_offset, double decay_bound) {
    return std::get<0>(reverb_forward(x, noise, fspec, gains, decays, mix, L, taps, nb, seed, seed_offset, decay_bound, false));
}
struct ReverbFn : public torch::autograd::Function<ReverbFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const c10::optional<Tensor>& noise, const Tensor& fspec, const Tensor& gains, const Tensor& decays,
                          const Tensor& mix, int64_t L, int64_t taps, int64_t nb, int64_t seed, const c10::optional<Tensor>& seed_offset, double decay_bound) {
        TORCH_CHECK(!(noise.has_value() && noise->defined() && noise->requires_grad()) && !fspec.requires_grad(),
                    "noise_shaped_reverberation: `noise` and the filters are not differentiable inputs (detach them)");
        const bool need = x.requires_grad() || gains.requires_grad() || decays.requires_grad() || mix.requires_grad();
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_reverb_forward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const c10::optional<Tensor>&, const Tensor&, const Tensor&, const Tensor&,
                                                                               const Tensor&, int64_t, int64_t, int64_t, int64_t, const c10::optional<Tensor>&, double, bool)>();
        auto [y, A, H, ir] = op.call(x, noise, fspec, gains, decays, mix, L, taps, nb, seed, seed_offset, decay_bound, need);
        if (need) {
            const bool has_noise = noise.has_value() && noise->defined(), has_off = seed_offset.has_value() && seed_offset->defined();
            // the seed-offset word is read again by the adjoint kernels when they run: saved WITH the tensors, so that an in-place bump between
            // forward and backward trips autograd's version check instead of regenerating different noise
            ctx->save_for_backward({ir, A, H, fspec, gains, decays, mix, has_noise ? *noise : Tensor(), has_off ? *seed_offset : Tensor()});
            ctx->saved_data["Cx"] = x.size(1); ctx->saved_data["L"] = L; ctx->saved_data["taps"] = taps; ctx->saved_data["nb"] = nb; ctx->saved_data["seed"] = seed;
            ctx->saved_data["bound"] = decay_bound;
            ctx->saved_data["has_noise"] = has_noise;
            ctx->saved_data["dt"] = std::vector<int64_t>{(int64_t)gains.scalar_type(), (int64_t)decays.scalar_type(), (int64_t)mix.scalar_type()};
            ctx->saved_data["mix_shape"] = mix.sizes().vec();
        }
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto s = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_reverb_backward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const c10::optional<Tensor>&,
                                                                               const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, int64_t, int64_t,
                                                                               int64_t, const c10::optional<Tensor>&, double)>();
        const c10::optional<Tensor> noise = s[7].defined() ? c10::optional<Tensor>(s[7]) : c10::nullopt;
        const c10::optional<Tensor> off = s[8].defined() ? c10::optional<Tensor>(s[8]) : c10::nullopt;
        auto [gx, gg, gd, gm] = op.call(grads[0], s[0], s[1], s[2], noise, s[3], s[4], s[5], s[6], ctx->saved_data["Cx"].toInt(), ctx->saved_data["L"].toInt(),
                                        ctx->saved_data["taps"].toInt(), ctx->saved_data["nb"].toInt(), ctx->saved_data["seed"].toInt(), off,
                                        ctx->saved_data["bound"].toDouble());
        const auto dt = ctx->saved_data["dt"].toIntVector();
        // needs_input_grad counts the TENSOR arguments that were passed (an absent optional has no edge): x, [noise], fspec, gains, decays, mix
        const size_t e = ctx->saved_data["has_noise"].toBool() ? 1 : 0;
        return {ctx->needs_input_grad(0) ? gx : Tensor(), Tensor(), Tensor(), ctx->needs_input_grad(2 + e) ? gg.to((at::ScalarType)dt[0]) : Tensor(),
                ctx->needs_input_grad(3 + e) ? gd.to((at::ScalarType)dt[1]) : Tensor(),
                ctx->needs_input_grad(4 + e) ? gm.to((at::ScalarType)dt[2]).reshape(ctx->saved_data["mix_shape"].toIntVector()) : Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};
Tensor reverb_autograd(const Tensor& x, const c10::optional<Tensor>& noise, const Tensor& fspec, const Tensor& gains, const Tensor& decays, const Tensor& mix, int64_t L,
                       int64_t taps, int64_t nb, int64_t seed, const c10::optional<Tensor>& seed_offset, double decay_bound) {
    return ReverbFn::apply(x, noise, fspec, gains, decays, mix, L, taps, nb, seed, seed_offset, decay_bound);
}


// =====================================================================================================================================
// The reference's own callables (round 5): the signatures of dasp_pytorch.functional / .signal on float32 ROCm tensors.
// =====================================================================================================================================

// ---- functional.parametric_eq on its 3 S control tensors (functional.py:118-139; ops.ParametricEQFunction; dasp_peq_forward / _backward) ----
// controls[3 k + c]: control c (0 gain_db, 1 cutoff_freq, 2 q_factor) of section k, Bp = 1 or bs values each, any shape; they are read in
// place by the design kernel (a host array of device pointers: no packing copy) and their gradients come back as the rows of one (3 S, Bp)
// matrix - one contiguous row per control tensor, no stack node, no 3 S copy kernels.
struct PeqRows { int64_t B, C, N, Bp, S; };
PeqRows peq_rows_check(const Tensor& x, at::TensorList controls, at::IntArrayRef types) {
    need_device(x, "x");
    TORCH_CHECK(x.dim() == 3, "dasp::parametric_eq: x must be (bs, chs, seq_len), got ", x.sizes());
    TORCH_CHECK(x.scalar_type() == at::kFloat, "dasp::parametric_eq computes in float32; got ", x.scalar_type());
    const int64_t S = (int64_t)types.size();
    TORCH_CHECK(dasp_sos_supported_sections((int)S), "no kernel compiled for ", S, " sections");
    TORCH_CHECK((int64_t)controls.size() == 3 * S, "dasp::parametric_eq: ", 3 * S, " control tensors for ", S, " sections, got ", controls.size());
    const int64_t Bp = nel(controls[0]);
    for (const Tensor& c : controls) {
        same_device(x, c, "control");
        TORCH_CHECK(nel(c) == Bp && (Bp == 1 || Bp == x.size(0)), "parametric_eq controls must each hold ", x.size(0), " (or 1) values, got ", nel(c));
    }
    for (int64_t t : types) TORCH_CHECK(t >= 0 && t <= 4, "dasp::parametric_eq: filter type ", t, " (0 peaking, 1 low_shelf, 2 high_shelf, 3 low_pass, 4 high_pass)");
    return PeqRows{x.size(0), x.size(1), x.size(2), Bp, S};
}
std::tuple<Tensor, Tensor, Tensor> peq_forward(const Tensor& x, at::TensorList controls, double sample_rate, at::IntArrayRef types, int64_t tseg, bool save) {
    const PeqRows d = peq_rows_check(x, controls, types);
    c10::DeviceGuard guard(x.device());
    const Tensor x32 = x.contiguous();
    Tensor y = at::empty_like(x32);
    const long n_tab = round64(d.Bp * dasp_sos_table_floats((int)d.S));
    const long n_car = save ? round64(dasp_sos_carry_floats(d.B * d.C, d.N, (int)d.S)) : 0;
    const long n_dt = d.Bp * dasp_sos_dtab_doubles((int)d.S), n_st = tseg ? d.Bp * dasp_sos_segtab_doubles((int)d.S) : 0;
    Tensor work32 = empty_f32(n_tab + n_car, x32);
    Tensor work64 = at::empty({n_dt + n_st}, x32.options().dtype(at::kDouble));
    if (x32.numel() == 0) return {y, work32, work64};
    Tensor segbuf = tseg ? empty_f32(round64(dasp_sos_seg_floats(d.B * d.C, d.N, (int)d.S, tseg)), x32) : Tensor();
    std::vector<Tensor> keep;               // float32 contiguous views / copies of the controls, alive until the launch is queued
    std::vector<const float*> rows;
    keep.reserve(controls.size()); rows.reserve(controls.size());
    for (const Tensor& c : controls) {
        keep.push_back(c.scalar_type() == at::kFloat && c.is_contiguous() ? c : c.to(at::kFloat).contiguous());
        rows.push_back(keep.back().data_ptr<float>());
    }
    std::vector<int> ty(types.begin(), types.end());
    float* w = work32.data_ptr<float>();
    double* w64 = work64.data_ptr<double>();
    check_rc(dasp_peq_forward(rows.data(), (int)d.Bp, (int)d.S, ty.data(), sample_rate, w, w64, x32.data_ptr<float>(), y.data_ptr<float>(),
                              save ? w + n_tab : nullptr, (int)d.B, (int)d.C, d.N, tseg, tseg ? w64 + n_dt : nullptr, fp(segbuf), stream_of(x32)),
             "dasp_peq_forward");
    return {y, work32, work64};
}
std::tuple<Tensor, Tensor> peq_backward(const Tensor& x, const Tensor& gy, const Tensor& work32, const Tensor& work64, int64_t Bp, int64_t S,
                                        int64_t tseg, bool need_gx, bool need_gc) {
    return peq_backward_impl(x, gy, work32, work64, Bp, S, tseg, need_gx, need_gc, 2);
}
Tensor peq_device(const Tensor& x, double sample_rate, at::TensorList controls, at::IntArrayRef types) {
    const PeqRows d = peq_rows_check(x, controls, types);
    return std::get<0>(peq_forward(x, controls, sample_rate, types, sos_segment_tiles(d.B * d.C, d.N), false));
}
struct PeqFn : public torch::autograd::Function<PeqFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, double sample_rate, at::TensorList controls, std::vector<int64_t> types) {
        const PeqRows d = peq_rows_check(x, controls, types);
        const int64_t tseg = sos_segment_tiles(d.B * d.C, d.N);
        bool need = x.requires_grad();
        for (const Tensor& c : controls) need = need || c.requires_grad();
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_peq_forward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, at::TensorList, double, at::IntArrayRef, int64_t, bool)>();
        auto [y, w32, w64] = op.call(x, controls, sample_rate, types, tseg, need);
        if (need) {
            variable_list keep = {x, w32, w64};
            for (const Tensor& c : controls) keep.push_back(c);       // (a few values each: their gradients go back in their shape and dtype)
            ctx->save_for_backward(keep);
            ctx->saved_data["Bp"] = d.Bp; ctx->saved_data["S"] = d.S; ctx->saved_data["tseg"] = tseg;
        }
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_peq_backward", "")
                             .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, int64_t, bool, bool)>();
        const int64_t S = ctx->saved_data["S"].toInt(), nc = 3 * S;
        const bool need_gx = ctx->needs_input_grad(0);
        bool need_gc = false;
        for (int64_t i = 0; i < nc; ++i) need_gc = need_gc || ctx->needs_input_grad(1 + i);
        auto [gx, gc] = op.call(saved[0], grads[0], saved[1], saved[2], ctx->saved_data["Bp"].toInt(), S, ctx->saved_data["tseg"].toInt(), need_gx, need_gc);
        // forward arguments: x, sample_rate, the 3 S controls, types
        variable_list out(2 + nc + 1);
        if (need_gx) out[0] = gx;
        if (need_gc) {
            for (int64_t i = 0; i < nc; ++i)
                if (ctx->needs_input_grad(1 + i)) out[2 + i] = like_control(gc.select(0, i), saved[3 + i]);
        }
        return out;
    }
};
Tensor peq_autograd(const Tensor& x, double sample_rate, at::TensorList controls, at::IntArrayRef types) {
    return PeqFn::apply(x, sample_rate, controls, types.vec());
}

// ---- functional.compressor / expander on their six control tensors (functional.py:275-286; ops.DynamicsFunction) ------------------------------
// The five controls the kernels read are stacked into the (bs, 5) rows of dasp::dynamics_ctl's entry points; release_ms has no path to the
// output (functional.py:340,343-344): it is accepted and gets a zero gradient.
struct Dyn6Fn : public torch::autograd::Function<Dyn6Fn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, double sample_rate, const Tensor& threshold_db, const Tensor& ratio, const Tensor& attack_ms,
                          const Tensor& release_ms, const Tensor& knee_db, const Tensor& makeup_gain_db, double eps, int64_t lookahead, int64_t mode) {
        need_device(x, "x");
        const Tensor* six[6] = {&threshold_db, &ratio, &attack_ms, &release_ms, &knee_db, &makeup_gain_db};
        bool need = x.requires_grad();
        for (const Tensor* c : six) {
            same_device(x, *c, "control");
            // the reference's .view(-1, 1, 1) against a (bs, 1, seq_len) side chain: no parameter broadcasting (functional.py:330-336)
            TORCH_CHECK(x.dim() == 3 && nel(*c) == x.size(0), "The size of tensor a (", nel(*c), ") must match the size of tensor b (", x.dim() ? x.size(0) : 0,
                        ") at non-singleton dimension 0");
            need = need || c->requires_grad();
        }
        at::AutoDispatchBelowADInplaceOrView below;
        // the five controls the kernels read, as contiguous float32 vectors (no-ops for float32 controls of bs values: nothing is launched)
        const std::vector<Tensor> five = {flat32(threshold_db).contiguous(), flat32(ratio).contiguous(), flat32(attack_ms).contiguous(),
                                          flat32(knee_db).contiguous(), flat32(makeup_gain_db).contiguous()};
        dyn6_check(x, five, mode, lookahead);
        const int64_t tseg = dyn_segment_tiles(x.size(0), x.size(2));
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_dynamics6_forward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, at::TensorList, int64_t, double, double, int64_t, int64_t, bool)>();
        auto [y, carries, lin] = op.call(x, five, mode, sample_rate, eps, lookahead, tseg, need);
        if (need) {
            ctx->save_for_backward({x, carries, lin, five[0], five[1], five[2], five[3], five[4], threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db});
            ctx->saved_data["mode"] = mode; ctx->saved_data["sr"] = sample_rate; ctx->saved_data["eps"] = eps;
            ctx->saved_data["look"] = lookahead; ctx->saved_data["tseg"] = tseg;
        }
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto s = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_dynamics6_backward", "")
                             .typed<std::tuple<Tensor, Tensor>(const Tensor&, at::TensorList, const Tensor&, const Tensor&, const Tensor&, int64_t, double, double,
                                                               int64_t, int64_t)>();
        const std::vector<Tensor> five = {s[3], s[4], s[5], s[6], s[7]};
        auto [gx, g6] = op.call(s[0], five, grads[0], s[1], s[2], ctx->saved_data["mode"].toInt(), ctx->saved_data["sr"].toDouble(),
                                ctx->saved_data["eps"].toDouble(), ctx->saved_data["look"].toInt(), ctx->saved_data["tseg"].toInt());
        // forward arguments: x, sample_rate, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db, eps, lookahead, mode; row i
        // of g6 is the gradient of control i in that order (release_ms: zeros), a contiguous vector each: AccumulateGrad takes them as they are
        variable_list out(11);
        if (ctx->needs_input_grad(0)) out[0] = gx;
        for (int i = 0; i < 6; ++i)
            if (ctx->needs_input_grad(1 + i)) out[2 + i] = like_control(g6.select(0, i), s[8 + i]);       // (needs_input_grad counts tensor inputs only)
        return out;
    }
};
Tensor dyn6_device(const Tensor& x, double sample_rate, const Tensor& threshold_db, const Tensor& ratio, const Tensor& attack_ms, const Tensor& release_ms,
                   const Tensor& knee_db, const Tensor& makeup_gain_db, double eps, int64_t lookahead, int64_t mode) {
    need_device(x, "x");
    for (const Tensor* c : {&threshold_db, &ratio, &attack_ms, &release_ms, &knee_db, &makeup_gain_db})
        TORCH_CHECK(x.dim() == 3 && c->numel() == x.size(0), "The size of tensor a (", c->numel(), ") must match the size of tensor b (", x.dim() ? x.size(0) : 0,
                    ") at non-singleton dimension 0");
    const std::vector<Tensor> five = {flat32(threshold_db).contiguous(), flat32(ratio).contiguous(), flat32(attack_ms).contiguous(), flat32(knee_db).contiguous(),
                                      flat32(makeup_gain_db).contiguous()};
    return std::get<0>(dyn6_forward(x, five, mode, sample_rate, eps, lookahead, dyn_segment_tiles(x.size(0), x.size(2)), false));
}
Tensor dyn6_autograd(const Tensor& x, double sample_rate, const Tensor& threshold_db, const Tensor& ratio, const Tensor& attack_ms, const Tensor& release_ms,
                     const Tensor& knee_db, const Tensor& makeup_gain_db, double eps, int64_t lookahead, int64_t mode) {
    return Dyn6Fn::apply(x, sample_rate, threshold_db, ratio, attack_ms, release_ms, knee_db, makeup_gain_db, eps, lookahead, mode);
}

// ---- functional.gain / functional.distortion (functional.py:10-29, :65-78; ops.GainFunction / DistortionFunction) ---------------------------------
// op 0: y = x 10^(gain_db / 20), one value per batch item; op 1: y = tanh(x 10^(drive_db / 20)), one value per (item, channel) row
void ew_check(const Tensor& x, const Tensor& ctl, int64_t op) {
    need_device(x, "x");
    same_device(x, ctl, op == 0 ? "gain_db" : "drive_db");
    TORCH_CHECK(op == 0 || op == 1, "dasp::_ew_forward: op 0 (gain) or 1 (distortion)");
    TORCH_CHECK(x.dim() == 3 && x.scalar_type() == at::kFloat, "dasp::", op == 0 ? "gain" : "distortion", ": x must be float32 (bs, chs, seq_len), got ", x.scalar_type(), " ", x.sizes());
    const int64_t want = op == 0 ? x.size(0) : x.size(0) * x.size(1);
    if (op == 0) {
        TORCH_CHECK(nel(ctl) == want, "shape '[", x.size(0), ", 1, 1]' is invalid for input of size ", nel(ctl));
    } else {
        TORCH_CHECK(nel(ctl) == want, "shape '[", x.size(0), ", ", x.size(1), ", -1]' is invalid for input of size ", nel(ctl));
    }
}
Tensor ew_forward(const Tensor& x, const Tensor& ctl, int64_t op) {
    ew_check(x, ctl, op);
    c10::DeviceGuard guard(x.device());
    const Tensor x32 = x.contiguous(), c32 = f32c(ctl.reshape({-1}));
    Tensor y = at::empty_like(x32);
    if (x32.numel() == 0) return y;
    const int B = (int)x.size(0), C = (int)x.size(1);
    const long N = x.size(2);
    check_rc(op == 0 ? dasp_gain_forward(x32.data_ptr<float>(), c32.data_ptr<float>(), y.data_ptr<float>(), B, C, N, stream_of(x32))
                     : dasp_distortion_forward(x32.data_ptr<float>(), c32.data_ptr<float>(), y.data_ptr<float>(), B, C, N, stream_of(x32)),
             op == 0 ? "dasp_gain_forward" : "dasp_distortion_forward");
    return y;
}
std::tuple<Tensor, Tensor> ew_backward(const Tensor& x, const Tensor& ctl, const Tensor& gy, int64_t op) {
    ew_check(x, ctl, op);
    same_device(x, gy, "grad_output");
    TORCH_CHECK(gy.sizes() == x.sizes(), "dasp::_ew_backward: grad_output ", gy.sizes(), " does not match x ", x.sizes());
    c10::DeviceGuard guard(x.device());
    const Tensor x32 = x.contiguous(), c32 = f32c(ctl.reshape({-1})), g32 = f32c(gy);
    Tensor gx = at::empty_like(x32), gctl = at::empty_like(c32);
    if (x32.numel() == 0) return {gx, gctl.zero_()};
    const int B = (int)x.size(0), C = (int)x.size(1);
    const long N = x.size(2);
    Tensor partials = empty_f32(dasp_ew_partial_floats((long)B * C, N), x32);
    check_rc(op == 0 ? dasp_gain_backward(x32.data_ptr<float>(), c32.data_ptr<float>(), g32.data_ptr<float>(), gx.data_ptr<float>(), gctl.data_ptr<float>(),
                                          partials.data_ptr<float>(), B, C, N, stream_of(x32))
                     : dasp_distortion_backward(x32.data_ptr<float>(), c32.data_ptr<float>(), g32.data_ptr<float>(), gx.data_ptr<float>(), gctl.data_ptr<float>(),
                                                partials.data_ptr<float>(), B, C, N, stream_of(x32)),
             op == 0 ? "dasp_gain_backward" : "dasp_distortion_backward");
    return {gx, gctl};
}
struct EwFn : public torch::autograd::Function<EwFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& ctl, int64_t op) {
        ew_check(x, ctl, op);
        at::AutoDispatchBelowADInplaceOrView below;
        static auto fop = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_ew_forward", "").typed<Tensor(const Tensor&, const Tensor&, int64_t)>();
        Tensor y = fop.call(x, ctl, op);
        if (x.requires_grad() || ctl.requires_grad()) {
            ctx->save_for_backward({x, ctl});
            ctx->saved_data["op"] = op;
        }
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto s = ctx->get_saved_variables();
        static auto bop = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_ew_backward", "")
                              .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, int64_t)>();
        auto [gx, gctl] = bop.call(s[0], s[1], grads[0], ctx->saved_data["op"].toInt());
        return {ctx->needs_input_grad(0) ? gx : Tensor(), ctx->needs_input_grad(1) ? gctl.reshape(s[1].sizes()).to(s[1].scalar_type()) : Tensor(), Tensor()};
    }
};
Tensor gain_device(const Tensor& x, const Tensor& gain_db) { return ew_forward(x, gain_db, 0); }
Tensor gain_autograd(const Tensor& x, const Tensor& gain_db) { return EwFn::apply(x, gain_db, (int64_t)0); }
Tensor distortion_device(const Tensor& x, const Tensor& drive_db) { return ew_forward(x, drive_db, 1); }
Tensor distortion_autograd(const Tensor& x, const Tensor& drive_db) { return EwFn::apply(x, drive_db, (int64_t)1); }

// ---- signal.sosfilt_via_fsm on at most 8 sections (signal.py:136-166; ops.SosFiltFunction) -----------------------------------------------
// sos (Bs, S, 6) rows [b0 b1 b2 a0 a1 a2], Bs = 1 or bs; x (bs, chs, seq_len). S is padded to a compiled section count with identity
// sections; longer cascades are the caller's successive calls (dasp_pytorch_amd.signal).
struct SosDims { int64_t B, C, N, Bs, S, Sp; };
SosDims sos_check(const Tensor& sos, const Tensor& x) {
    need_device(x, "x");
    same_device(x, sos, "sos");
    TORCH_CHECK(x.dim() == 3 && x.scalar_type() == at::kFloat, "dasp::sosfilt: x must be float32 (bs, chs, seq_len), got ", x.scalar_type(), " ", x.sizes());
    TORCH_CHECK(sos.dim() == 3 && sos.size(2) == 6, "dasp::sosfilt: sos must be (bs, n_sections, 6), got ", sos.sizes());
    const int64_t S = sos.size(1);
    TORCH_CHECK(S >= 1 && S <= 8, "more than 8 sections per call: chain calls (see signal.sosfilt_via_fsm)");
    TORCH_CHECK(sos.size(0) == 1 || sos.size(0) == x.size(0), "dasp::sosfilt: sos holds ", sos.size(0), " filter sets for a batch of ", x.size(0));
    return SosDims{x.size(0), x.size(1), x.size(2), sos.size(0), S, (S + 1) / 2 * 2};
}
std::tuple<Tensor, Tensor, Tensor> sosfilt_forward(const Tensor& sos, const Tensor& x, int64_t tseg, bool save) {
    const SosDims d = sos_check(sos, x);
    c10::DeviceGuard guard(x.device());
    const Tensor x32 = x.contiguous();
    Tensor y = at::empty_like(x32);
    const long n_tab = round64(d.Bs * dasp_sos_table_floats((int)d.Sp));
    const long n_car = save ? round64(dasp_sos_carry_floats(d.B * d.C, d.N, (int)d.Sp)) : 0;
    const long n_dt = d.Bs * dasp_sos_dtab_doubles((int)d.Sp), n_st = tseg ? d.Bs * dasp_sos_segtab_doubles((int)d.Sp) : 0;
    Tensor work32 = empty_f32(n_tab + n_car, x32);
    Tensor work64 = at::empty({n_dt + n_st}, x32.options().dtype(at::kDouble));
    if (x32.numel() == 0) return {y, work32, work64};
    Tensor s32 = f32c(sos);
    if (d.Sp != d.S) {                       // identity sections [1 0 0 1 0 0]
        Tensor pad = at::zeros({d.Bs, d.Sp - d.S, 6}, s32.options());
        pad.select(2, 0).fill_(1.0);
        pad.select(2, 3).fill_(1.0);
        s32 = at::cat({s32, pad}, 1).contiguous();
    }
    float* w = work32.data_ptr<float>();
    double* w64 = work64.data_ptr<double>();
    void* st = stream_of(x32);
    check_rc(dasp_sos_prepare(s32.data_ptr<float>(), (int)d.Bs, (int)d.Sp, w, w64, st), "dasp_sos_prepare");
    if (tseg) {
        Tensor segbuf = empty_f32(round64(dasp_sos_seg_floats(d.B * d.C, d.N, (int)d.Sp, tseg)), x32);
        check_rc(dasp_sos_segment_prepare(w64, (int)d.Bs, (int)d.Sp, tseg, w64 + n_dt, st), "dasp_sos_segment_prepare");
        check_rc(dasp_sosfilt_forward_seg(w, w64 + n_dt, (int)d.Bs, x32.data_ptr<float>(), y.data_ptr<float>(), save ? w + n_tab : nullptr, segbuf.data_ptr<float>(),
                                          (int)d.B, (int)d.C, d.N, (int)d.Sp, tseg, st),
                 "dasp_sosfilt_forward_seg");
    } else {
        check_rc(dasp_sosfilt_forward(w, (int)d.Bs, x32.data_ptr<float>(), y.data_ptr<float>(), save ? w + n_tab : nullptr, (int)d.B, (int)d.C, d.N, (int)d.Sp, st),
                 "dasp_sosfilt_forward");
    }
    return {y, work32, work64};
}
// -> (gx or empty, gsos (Bs, Sp, 6) or empty)
std::tuple<Tensor, Tensor> sosfilt_backward(const Tensor& x, const Tensor& gy, const Tensor& work32, const Tensor& work64, int64_t Bs, int64_t Sp, int64_t tseg,
                                            bool need_gx, bool need_gs) {
    need_device(x, "x");
    same_device(x, gy, "grad_output");
    TORCH_CHECK(gy.sizes() == x.sizes(), "dasp::_sosfilt_backward: grad_output ", gy.sizes(), " does not match x ", x.sizes());
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), C = x.size(1), N = x.size(2);
    const Tensor x32 = x.contiguous(), g32 = f32c(gy);
    Tensor gx = need_gx ? at::empty_like(x32) : at::empty({0}, x32.options());
    Tensor gs = need_gs ? at::empty({B, Sp, 6}, x32.options()) : at::empty({0}, x32.options());
    if (x32.numel() == 0 || (!need_gx && !need_gs)) return {gx, need_gs ? at::zeros({Bs, Sp, 6}, x32.options()) : gs};
    const long n_tab = round64(Bs * dasp_sos_table_floats((int)Sp)), n_dt = Bs * dasp_sos_dtab_doubles((int)Sp);
    const long G = dasp_sos_segments(N, tseg);
    TORCH_CHECK(work32.numel() >= n_tab + round64(dasp_sos_carry_floats(B * C, N, (int)Sp)) && work64.numel() >= n_dt + (tseg ? Bs * dasp_sos_segtab_doubles((int)Sp) : 0),
                "dasp::_sosfilt_backward: work buffers do not belong to a forward call of this shape");
    Tensor partials = need_gs ? empty_f32(round64(dasp_sos_partial_floats(B * C * G, (int)Sp)), x32) : Tensor();
    float* w = work32.data_ptr<float>();
    double* w64 = work64.data_ptr<double>();
    void* st = stream_of(x32);
    if (tseg) {
        Tensor segbuf = empty_f32(round64(dasp_sos_seg_floats(B * C, N, (int)Sp, tseg)), x32);
        check_rc(dasp_sosfilt_backward_seg_ex(w, w64 + n_dt, (int)Bs, x32.data_ptr<float>(), g32.data_ptr<float>(), w + n_tab, need_gx ? gx.data_ptr<float>() : nullptr,
                                              fp(partials), segbuf.data_ptr<float>(), (int)B, (int)C, N, (int)Sp, tseg, st),
                 "dasp_sosfilt_backward_seg_ex");
        if (need_gs) check_rc(dasp_sos_grad_finalize_ex(w64, (int)Bs, partials.data_ptr<float>(), (int)B, (int)C, (int)Sp, (int)G, 0, gs.data_ptr<float>(), st), "dasp_sos_grad_finalize_ex");
    } else {
        check_rc(dasp_sosfilt_backward_grads_ex(w, w64, (int)Bs, x32.data_ptr<float>(), g32.data_ptr<float>(), w + n_tab, need_gx ? gx.data_ptr<float>() : nullptr, fp(partials), 0,
                                                need_gs ? gs.data_ptr<float>() : nullptr, (int)B, (int)C, N, (int)Sp, st),
                 "dasp_sosfilt_backward_grads_ex");
    }
    if (need_gs && Bs == 1 && B != 1) gs = gs.sum(0, /*keepdim=*/true);
    return {gx, gs};
}
Tensor sosfilt_device(const Tensor& sos, const Tensor& x) {
    const SosDims d = sos_check(sos, x);
    return std::get<0>(sosfilt_forward(sos, x, sos_segment_tiles(d.B * d.C, d.N, true), false));
}
struct SosFn : public torch::autograd::Function<SosFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& sos, const Tensor& x) {
        const SosDims d = sos_check(sos, x);
        const int64_t tseg = sos_segment_tiles(d.B * d.C, d.N, true);
        const bool need = sos.requires_grad() || x.requires_grad();
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_sosfilt_forward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, int64_t, bool)>();
        auto [y, w32, w64] = op.call(sos, x, tseg, need);
        if (need) {
            ctx->save_for_backward({x, w32, w64});
            ctx->saved_data["Bs"] = d.Bs; ctx->saved_data["S"] = d.S; ctx->saved_data["Sp"] = d.Sp; ctx->saved_data["tseg"] = tseg;
            ctx->saved_data["dt"] = (int64_t)sos.scalar_type();
        }
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto s = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_sosfilt_backward", "")
                             .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, int64_t, bool, bool)>();
        const bool need_gs = ctx->needs_input_grad(0), need_gx = ctx->needs_input_grad(1);
        auto [gx, gs] = op.call(s[0], grads[0], s[1], s[2], ctx->saved_data["Bs"].toInt(), ctx->saved_data["Sp"].toInt(), ctx->saved_data["tseg"].toInt(), need_gx, need_gs);
        return {need_gs ? gs.narrow(1, 0, ctx->saved_data["S"].toInt()).to((at::ScalarType)ctx->saved_data["dt"].toInt()) : Tensor(), need_gx ? gx : Tensor()};
    }
};
Tensor sosfilt_autograd(const Tensor& sos, const Tensor& x) { return SosFn::apply(sos, x); }

// ---- functional.noise_shaped_reverberation on its 12 + 12 + 1 control tensors (functional.py:406-436) -----------------------------------------
// The band gains and decays are stacked into the (bs, bands) matrices of dasp::reverb's entry points, their gradients go back as the rows of
// the transposed gradient matrices (one contiguous row per control; autograd's own stack backward hands out 24 strided columns, each of
// which AccumulateGrad copies with a kernel of its own).
struct NsrFn : public torch::autograd::Function<NsrFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, at::TensorList band_gains, at::TensorList band_decays, const Tensor& mix, const c10::optional<Tensor>& noise,
                          const Tensor& fspec, int64_t L, int64_t taps, int64_t seed, const c10::optional<Tensor>& seed_offset, double decay_bound) {
        need_device(x, "x");
        const int64_t nb = (int64_t)band_gains.size(), B = x.dim() ? x.size(0) : 0;
        TORCH_CHECK(nb >= 1 && (int64_t)band_decays.size() == nb, "dasp::noise_shaped_reverb: as many band decays as band gains");
        TORCH_CHECK(!(noise.has_value() && noise->defined() && noise->requires_grad()) && !fspec.requires_grad(),
                    "noise_shaped_reverberation: `noise` and the filters are not differentiable inputs (detach them)");
        bool need = x.requires_grad() || mix.requires_grad();
        std::vector<Tensor> gv, dv;
        variable_list ctls;
        auto note = [&](const Tensor& c) {
            same_device(x, c, "control");
            // the reference's torch.stack(...).view(bs, 12) / mix.view(bs, 1, 1) (functional.py:498-544): no broadcasting
            TORCH_CHECK(nel(c) == B, "shape '[", B, ", ", nb, "]' is invalid for input of size ", nb * nel(c));
            need = need || c.requires_grad();
            ctls.push_back(c);
        };
        for (const Tensor& c : band_gains) note(c);
        for (const Tensor& c : band_decays) note(c);
        note(mix);
        at::AutoDispatchBelowADInplaceOrView below;
        for (const Tensor& c : band_gains) gv.push_back(flat32(c));
        for (const Tensor& c : band_decays) dv.push_back(flat32(c));
        const Tensor gains = at::stack(gv, 1), decays = at::stack(dv, 1), m32 = flat32(mix);
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_reverb_forward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const c10::optional<Tensor>&, const Tensor&, const Tensor&, const Tensor&,
                                                                               const Tensor&, int64_t, int64_t, int64_t, int64_t, const c10::optional<Tensor>&, double, bool)>();
        auto [y, A, H, ir] = op.call(x, noise, fspec, gains, decays, m32, L, taps, nb, seed, seed_offset, decay_bound, need);
        if (need) {
            const bool has_noise = noise.has_value() && noise->defined(), has_off = seed_offset.has_value() && seed_offset->defined();
            variable_list keep = {ir, A, H, fspec, gains, decays, m32, has_noise ? *noise : Tensor(), has_off ? *seed_offset : Tensor()};
            keep.insert(keep.end(), ctls.begin(), ctls.end());       // (a few values each: their gradients go back in their shape and dtype)
            ctx->save_for_backward(keep);
            ctx->saved_data["Cx"] = x.size(1); ctx->saved_data["L"] = L; ctx->saved_data["taps"] = taps; ctx->saved_data["nb"] = nb; ctx->saved_data["seed"] = seed;
            ctx->saved_data["bound"] = decay_bound; ctx->saved_data["has_noise"] = has_noise;
        }
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto s = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("dasp::_reverb_backward", "")
                             .typed<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const c10::optional<Tensor>&,
                                                                               const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, int64_t, int64_t,
                                                                               int64_t, const c10::optional<Tensor>&, double)>();
        const c10::optional<Tensor> noise = s[7].defined() ? c10::optional<Tensor>(s[7]) : c10::nullopt;
        const c10::optional<Tensor> off = s[8].defined() ? c10::optional<Tensor>(s[8]) : c10::nullopt;
        const int64_t nb = ctx->saved_data["nb"].toInt();
        auto [gx, gg, gd, gm] = op.call(grads[0], s[0], s[1], s[2], noise, s[3], s[4], s[5], s[6], ctx->saved_data["Cx"].toInt(), ctx->saved_data["L"].toInt(),
                                        ctx->saved_data["taps"].toInt(), nb, ctx->saved_data["seed"].toInt(), off, ctx->saved_data["bound"].toDouble());
        const Tensor ggr = gg.t().contiguous(), gdr = gd.t().contiguous();       // (bands, bs): one contiguous row per control tensor
        // forward arguments: x, the band gains, the band decays, mix, [noise], fspec, num_samples, taps, seed, [seed_offset], decay_bound.
        // needs_input_grad counts the tensor arguments that were passed: x, 2 bands controls, mix, ...
        variable_list out(1 + 2 * nb + 1 + 7);
        if (ctx->needs_input_grad(0)) out[0] = gx;
        for (int64_t i = 0; i < 2 * nb + 1; ++i) {
            if (!ctx->needs_input_grad(1 + i)) continue;
            const Tensor g = i < nb ? ggr.select(0, i) : (i < 2 * nb ? gdr.select(0, i - nb) : gm);
            out[1 + i] = like_control(g, s[9 + i]);
        }
        return out;
    }
};
Tensor nsr_device(const Tensor& x, at::TensorList band_gains, at::TensorList band_decays, const Tensor& mix, const c10::optional<Tensor>& noise, const Tensor& fspec, int64_t L,
                  int64_t taps, int64_t seed, const c10::optional<Tensor>& seed_offset, double decay_bound) {
    need_device(x, "x");
    const int64_t nb = (int64_t)band_gains.size(), B = x.dim() ? x.size(0) : 0;
    TORCH_CHECK(nb >= 1 && (int64_t)band_decays.size() == nb, "dasp::noise_shaped_reverb: as many band decays as band gains");
    std::vector<Tensor> gv, dv;
    for (const Tensor& c : band_gains) { TORCH_CHECK(c.numel() == B, "shape '[", B, ", ", nb, "]' is invalid for input of size ", nb * c.numel()); gv.push_back(flat32(c)); }
    for (const Tensor& c : band_decays) { TORCH_CHECK(c.numel() == B, "shape '[", B, ", ", nb, "]' is invalid for input of size ", nb * c.numel()); dv.push_back(flat32(c)); }
    TORCH_CHECK(mix.numel() == B, "shape '[", B, ", ", nb, "]' is invalid for input of size ", nb * mix.numel());
    return reverb_device(x, noise, fspec, at::stack(gv, 1), at::stack(dv, 1), flat32(mix), L, taps, nb, seed, seed_offset, decay_bound);
}
Tensor nsr_autograd(const Tensor& x, at::TensorList band_gains, at::TensorList band_decays, const Tensor& mix, const c10::optional<Tensor>& noise, const Tensor& fspec, int64_t L,
                    int64_t taps, int64_t seed, const c10::optional<Tensor>& seed_offset, double decay_bound) {
    return NsrFn::apply(x, band_gains, band_decays, mix, noise, fspec, L, taps, seed, seed_offset, decay_bound);
}

}  // namespace

#ifndef DASP_ABI_HASH
#define DASP_ABI_HASH 0
#endif
// the hash of include/dasp_hip.h this extension was compiled against (csrc/build.py); libdasp_hip.so carries its own (dasp_abi_hash):
// dasp_pytorch_amd._torch_ops refuses an extension whose hash differs from the kernel library's
int64_t abi_hash() { return (int64_t)DASP_ABI_HASH; }

TORCH_LIBRARY(dasp, m) {
    // ---- the reference's own callables (dasp_pytorch/functional.py:10, :65, :118-139, :275-286, :406-436; signal.py:136), differentiable ----
    m.def("parametric_eq(Tensor x, float sample_rate, Tensor[] controls, int[] types) -> Tensor");
    m.def("dynamics(Tensor x, float sample_rate, Tensor threshold_db, Tensor ratio, Tensor attack_ms, Tensor release_ms, Tensor knee_db, Tensor makeup_gain_db, "
          "float eps, int lookahead_samples, int mode) -> Tensor");
    m.def("gain(Tensor x, Tensor gain_db) -> Tensor");
    m.def("distortion(Tensor x, Tensor drive_db) -> Tensor");
    m.def("sosfilt(Tensor sos, Tensor x) -> Tensor");
    m.def("noise_shaped_reverb(Tensor x, Tensor[] band_gains, Tensor[] band_decays, Tensor mix, Tensor? noise, Tensor fspec, int num_samples, int taps, int seed, "
          "Tensor? seed_offset, float decay_bound) -> Tensor");
    // ---- the chain on normalised parameters (Processor.process_normalized of ParametricEQ, dasp_pytorch/modules.py:124-156 + functional.py:118-272;
    // compressor / expander on control rows; the reverb on control matrices), differentiable. range_flag: see flag_ptr above ----
    m.def("parametric_eq_norm(Tensor x, Tensor param_tensor, float sample_rate, int[] types, float[] lo, float[] span, Tensor(a!)? range_flag=None) -> Tensor");
    m.def("dynamics_ctl(Tensor x, Tensor ctl, int mode, float sample_rate, float eps, int lookahead_samples) -> Tensor");
    m.def("eq_dyn_norm(Tensor x, Tensor param_tensor, float sample_rate, int[] types, float[] lo, float[] span, Tensor ctl, int mode, float eps, "
          "Tensor(a!)? range_flag=None) -> Tensor");
    m.def("_eq_dyn_norm_forward(Tensor x, Tensor param_tensor, float sample_rate, int[] types, float[] lo, float[] span, Tensor ctl, int mode, float eps, "
          "Tensor(a!)? range_flag=None) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("chain_controls(Tensor comp_params, Tensor reverb_params, Tensor gain_params, float[] lo, float[] span, Tensor(a!)? range_flag=None) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("reverb(Tensor x, Tensor? noise, Tensor fspec, Tensor gains, Tensor decays, Tensor mix, int num_samples, int taps, int bands, int seed, Tensor? seed_offset, "
          "float decay_bound) -> Tensor");
    // the two directions as plain functional ops (traced by AOTAutograd; fake implementations: dasp_pytorch_amd/_torch_ops.py)
    m.def("_peq_forward(Tensor x, Tensor[] controls, float sample_rate, int[] types, int tseg, bool save) -> (Tensor, Tensor, Tensor)");
    m.def("_peq_backward(Tensor x, Tensor grad_y, Tensor work32, Tensor work64, int Bp, int S, int tseg, bool need_gx, bool need_gc) -> (Tensor, Tensor)");
    m.def("_ew_forward(Tensor x, Tensor ctl, int op) -> Tensor");
    m.def("_ew_backward(Tensor x, Tensor ctl, Tensor grad_y, int op) -> (Tensor, Tensor)");
    m.def("_sosfilt_forward(Tensor sos, Tensor x, int tseg, bool save) -> (Tensor, Tensor, Tensor)");
    m.def("_sosfilt_backward(Tensor x, Tensor grad_y, Tensor work32, Tensor work64, int Bs, int Sp, int tseg, bool need_gx, bool need_gs) -> (Tensor, Tensor)");
    m.def("_peq_norm_forward(Tensor x, Tensor param_tensor, float sample_rate, int[] types, float[] lo, float[] span, int tseg, bool save, Tensor(a!)? range_flag=None) "
          "-> (Tensor, Tensor, Tensor)");
    m.def("_peq_norm_backward(Tensor x, Tensor grad_y, Tensor work32, Tensor work64, int Bp, int S, int tseg, bool need_gx, bool need_gp) -> (Tensor, Tensor)");
    m.def("_dynamics_forward(Tensor x, Tensor ctl, int mode, float sample_rate, float eps, int lookahead_samples, int tseg, bool save) -> (Tensor, Tensor, Tensor)");
    m.def("_dynamics6_forward(Tensor x, Tensor[] five, int mode, float sample_rate, float eps, int lookahead_samples, int tseg, bool save) -> (Tensor, Tensor, Tensor)");
    m.def("_dynamics6_backward(Tensor x, Tensor[] five, Tensor grad_y, Tensor carries, Tensor lin, int mode, float sample_rate, float eps, int lookahead_samples, int tseg) "
          "-> (Tensor, Tensor)");
    m.def("_dynamics_backward(Tensor x, Tensor ctl, Tensor grad_y, Tensor carries, Tensor lin, int mode, float sample_rate, float eps, int lookahead_samples, int tseg) "
          "-> (Tensor, Tensor)");
    m.def("_chain_controls_backward(Tensor gctl, Tensor ggain, Tensor gdecay, Tensor gmix, float[] span) -> (Tensor, Tensor, Tensor)");
    m.def("_reverb_forward(Tensor x, Tensor? noise, Tensor fspec, Tensor gains, Tensor decays, Tensor mix, int num_samples, int taps, int bands, int seed, "
          "Tensor? seed_offset, float decay_bound, bool save) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("_reverb_backward(Tensor grad_y, Tensor ir, Tensor A, Tensor H, Tensor? noise, Tensor fspec, Tensor gains, Tensor decays, Tensor mix, int Cx, int num_samples, "
          "int taps, int bands, int seed, Tensor? seed_offset, float decay_bound) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("_abi_hash() -> int", &abi_hash);
    m.def("_plan_override(int sos_tiles, int dyn_tiles) -> ()", &plan_override);
}
// ROCm devices carry the CUDA dispatch key in PyTorch-ROCm builds
TORCH_LIBRARY_IMPL(dasp, CUDA, m) {
    m.impl("parametric_eq", &peq_device);
    m.impl("dynamics", &dyn6_device);
    m.impl("gain", &gain_device);
    m.impl("distortion", &distortion_device);
    m.impl("sosfilt", &sosfilt_device);
    m.impl("noise_shaped_reverb", &nsr_device);
    m.impl("parametric_eq_norm", &peq_norm_device);
    m.impl("dynamics_ctl", &dyn_device);
    m.impl("eq_dyn_norm", &eq_dyn_norm_device);
    m.impl("_eq_dyn_norm_forward", &eq_dyn_norm_forward);
    m.impl("chain_controls", &chain_controls);
    m.impl("reverb", &reverb_device);
    m.impl("_peq_forward", &peq_forward);
    m.impl("_peq_backward", &peq_backward);
    m.impl("_ew_forward", &ew_forward);
    m.impl("_ew_backward", &ew_backward);
    m.impl("_sosfilt_forward", &sosfilt_forward);
    m.impl("_sosfilt_backward", &sosfilt_backward);
    m.impl("_peq_norm_forward", &peq_norm_forward);
    m.impl("_peq_norm_backward", &peq_norm_backward);
    m.impl("_dynamics_forward", &dyn_forward);
    m.impl("_dynamics6_forward", &dyn6_forward);
    m.impl("_dynamics6_backward", &dyn6_backward);
    m.impl("_dynamics_backward", &dyn_backward);
    m.impl("_chain_controls_backward", &chain_controls_backward);
    m.impl("_reverb_forward", &reverb_forward);
    m.impl("_reverb_backward", &reverb_backward);
}
TORCH_LIBRARY_IMPL(dasp, Autograd, m) {
    m.impl("parametric_eq", &peq_autograd);
    m.impl("dynamics", &dyn6_autograd);
    m.impl("gain", &gain_autograd);
    m.impl("distortion", &distortion_autograd);
    m.impl("sosfilt", &sosfilt_autograd);
    m.impl("noise_shaped_reverb", &nsr_autograd);
    m.impl("parametric_eq_norm", &peq_norm_autograd);
    m.impl("dynamics_ctl", &dyn_autograd);
    m.impl("eq_dyn_norm", &eq_dyn_norm_autograd);
    m.impl("chain_controls", &chain_controls_autograd);
    m.impl("reverb", &reverb_autograd);
    // the two directions themselves carry no derivative: backpropagating through them (a double backward, or calling `_forward` on tensors
    // that require a gradient) raises "derivative for dasp::... is not implemented" instead of treating the result as a constant
    for (const char* name : {"_peq_forward", "_peq_backward", "_ew_forward", "_ew_backward", "_sosfilt_forward", "_sosfilt_backward", "_peq_norm_forward",
                             "_peq_norm_backward", "_dynamics_forward", "_dynamics_backward", "_dynamics6_forward", "_dynamics6_backward", "_chain_controls_backward", "_reverb_forward", "_eq_dyn_norm_forward",
                             "_reverb_backward"})
        m.impl(name, torch::autograd::autogradNotImplementedFallback());
}
