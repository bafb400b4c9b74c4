// Host side of the sparse-voxel engine's per-layer calls as a PyTorch-ROCm C++ extension (host compiler only; no device
// code): at::Tensor in, output allocation through torch's caching allocator, raw pointers +
// c10::hip::getCurrentHIPStream() out to the C ABI of libminsu3d_hip.so (include/minsu3d_hip.h).  One call per layer and
// direction, ~250 of them per training step: through ctypes each costs the interpreter ~20 us of pointer marshalling,
// two or three torch.empty calls and a 20..35-argument foreign call (tools/host_profile.py); here it is one pybind call.
// minsu3d_amd/backend.py uses this module when it is built (MS3D_HOST_EXT=0 switches back to ctypes, which stays the
// loader of last resort: same library, same kernels, same results).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include <tuple>

#include "../../include/minsu3d_hip.h"

namespace {

inline ms3d_stream_t cur() { return (ms3d_stream_t)c10::hip::getCurrentHIPStream().stream(); }

inline void check(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed with code ", rc, " (libminsu3d_hip.so; there is no CPU fallback)");
}

using OptT = c10::optional<at::Tensor>;

inline const float *fptr(const OptT &t)
{
    if (!t.has_value() || !t->defined()) return nullptr;
    TORCH_CHECK(t->is_contiguous() && t->scalar_type() == at::kFloat, "expected a contiguous float32 tensor");
    return t->data_ptr<float>();
}
inline const int *iptr(const OptT &t)
{
    if (!t.has_value() || !t->defined()) return nullptr;
    TORCH_CHECK(t->is_contiguous() && t->scalar_type() == at::kInt, "expected a contiguous int32 tensor");
    return t->data_ptr<int>();
}

}  // namespace

// -> (y [vout, cout], stats [nparts, 2, cout] or None).  W undefined: wf_buf already holds the current weight images.
std::tuple<at::Tensor, OptT> conv_layer_forward(const at::Tensor &x, const OptT &W, const at::Tensor &nbr_fwd, int64_t vout,
                                                int64_t K, int64_t cin, int64_t cout, bool mirror_bwd, const OptT &pre_scale,
                                                const OptT &pre_shift, bool pre_relu, const OptT &residual, const OptT &bias,
                                                at::Tensor wf_buf, int64_t nparts /* 0: no statistics */, const OptT &pl_tile_start,
                                                const OptT &pl_entries, int64_t ev_start, int64_t ev_stop)
{
    TORCH_CHECK(x.is_cuda() && x.is_contiguous() && x.scalar_type() == at::kFloat, "x: contiguous float32 device tensor");
    at::Tensor y = at::empty({vout, cout}, x.options());
    OptT stats;
    if (nparts > 0) stats = at::empty({nparts, 2, cout}, x.options());
    check(ms3d_spconv_layer_forward(x.data_ptr<float>(), fptr(W), nbr_fwd.data_ptr<int>(), (int)vout, (int)K, (int)cin,
                                    (int)cout, mirror_bwd ? 1 : 0, fptr(pre_scale), fptr(pre_shift), pre_relu ? 1 : 0,
                                    fptr(residual), fptr(bias), wf_buf.data_ptr<float>(), y.data_ptr<float>(),
                                    stats.has_value() ? stats->data_ptr<float>() : nullptr, iptr(pl_tile_start),
                                    iptr(pl_entries), (void *)ev_start, (void *)ev_stop, cur()),
          "ms3d_spconv_layer_forward");
    return {y, stats};
}

// -> (dx or None, dgb [2, cin] or None, dW [K, cin, cout], slabs or None, slabs to reduce later, launch description).
// defer_floats > 0: the backward-weight slabs go into a fresh tensor of that many floats and are NOT reduced
// (backend.WgradQueue does it); defer_launch: a layer on the f32 table walk does not launch its backward-weight kernel
// either, the 128-byte description comes back instead (empty bytes: it was launched).
std::tuple<OptT, OptT, at::Tensor, OptT, int64_t, py::bytes> conv_layer_backward(
    const at::Tensor &x, const at::Tensor &dy, const at::Tensor &wf_buf, const at::Tensor &nbr_fwd, const at::Tensor &nbr_bwd,
    int64_t vin, int64_t vout, int64_t K, int64_t cin, int64_t cout, const OptT &scale, const OptT &shift, const OptT &mean,
    const OptT &invstd, bool relu, bool training, bool need_dx, const OptT &dx_add, at::Tensor ws, const OptT &ol_kt_start,
    const OptT &ol_entries, const OptT &pl_tile_start, const OptT &pl_entries, int64_t ev0, int64_t ev1, int64_t ev2,
    int64_t ev3, int64_t defer_floats, bool defer_launch)
{
    TORCH_CHECK(x.is_cuda() && x.is_contiguous() && dy.is_contiguous(), "x / dy: contiguous device tensors");
    const bool has_bn = scale.has_value() && scale->defined();
    OptT dx, dgb, slabs;
    if (need_dx || has_bn) dx = at::empty({vin, cin}, x.options());
    if (has_bn) dgb = at::empty({2, cin}, x.options());
    at::Tensor dW = at::empty({K, cin, cout}, x.options());
    int nblk = 0;
    alignas(16) unsigned char launch[128];
    reinterpret_cast<int *>(launch)[4] = 0;
    if (defer_floats > 0) slabs = at::empty({defer_floats}, x.options());
    check(ms3d_spconv_layer_backward(
              x.data_ptr<float>(), dy.data_ptr<float>(), wf_buf.data_ptr<float>(), nbr_fwd.data_ptr<int>(),
              nbr_bwd.data_ptr<int>(), (int)vin, (int)vout, (int)K, (int)cin, (int)cout, fptr(scale), fptr(shift), fptr(mean),
              fptr(invstd), (has_bn && relu) ? 1 : 0, (has_bn && training) ? 1 : 0, need_dx ? 1 : 0,
              dx.has_value() ? dx->data_ptr<float>() : nullptr, need_dx ? fptr(dx_add) : nullptr,
              dgb.has_value() ? dgb->data_ptr<float>() : nullptr, dW.data_ptr<float>(), (float *)ws.data_ptr(), iptr(ol_kt_start),
              iptr(ol_entries), iptr(pl_tile_start), iptr(pl_entries), (void *)ev0, (void *)ev1, (void *)ev2, (void *)ev3,
              nullptr, nullptr, 0, slabs.has_value() ? slabs->data_ptr<float>() : nullptr,
              slabs.has_value() ? &nblk : nullptr, (slabs.has_value() && defer_launch) ? (void *)launch : nullptr, cur()),
          "ms3d_spconv_layer_backward");
    if (!need_dx) dx = c10::nullopt;
    const bool described = reinterpret_cast<int *>(launch)[4] != 0;
    return {dx, dgb, dW, slabs, (int64_t)nblk, py::bytes(reinterpret_cast<const char *>(launch), described ? 128 : 0)};
}

// batch statistics from a convolution epilogue's partials -> [4, C] = (mean, invstd, scale, shift); running stats updated
at::Tensor bn_finalize(const at::Tensor &partial, int64_t V, double eps, double momentum, const OptT &gamma, const OptT &beta,
                       OptT running_mean, OptT running_var)
{
    const int64_t C = partial.size(2);
    at::Tensor outs = at::empty({4, C}, partial.options());
    float *o = outs.data_ptr<float>();
    check(ms3d_bn_finalize(partial.data_ptr<float>(), (int)partial.size(0), (long)V, (int)C, (float)eps, (float)momentum,
                           fptr(gamma), fptr(beta),
                           running_mean.has_value() && running_mean->defined() ? running_mean->data_ptr<float>() : nullptr,
                           running_var.has_value() && running_var->defined() ? running_var->data_ptr<float>() : nullptr, o,
                           o + C, o + 2 * C, o + 3 * C, cur()),
          "ms3d_bn_finalize");
    return outs;
}

// An identity-skip residual block's forward in training mode -- bn_finalize, fused convolution, bn_finalize, fused convolution
// with the residual in its epilogue -- from ONE call (round 6): the same four library calls functional.ResBlockFn makes, in the
// same order on the same stream, minus three trips through the interpreter and the backend's per-call bookkeeping (the
// proposal network issues ~145 launches with the GPU waiting for each: interpreter time is what that stretch of a step
// costs).  Both weight buffers must already hold the current images (prepare_conv_weights).
// -> (y1 [V, C], y2 [V, C], stats of y2 [nparts, 2, C] or None, (mean, invstd, scale, shift) of bn0 [4, C], of bn1 [4, C])
std::tuple<at::Tensor, at::Tensor, OptT, at::Tensor, at::Tensor> res_block_forward(
    const at::Tensor &x, const at::Tensor &stats_in, at::Tensor wf1, at::Tensor wf2, const at::Tensor &nbr, int64_t V, int64_t C,
    const OptT &pl_tile_start, const OptT &pl_entries, int64_t nparts, bool want_stats, const OptT &g0, const OptT &b0,
    OptT rm0, OptT rv0, double eps0, double mom0, const OptT &g1, const OptT &b1, OptT rm1, OptT rv1, double eps1, double mom1)
{
    TORCH_CHECK(x.is_cuda() && x.is_contiguous() && x.scalar_type() == at::kFloat && x.size(1) == C, "x: contiguous f32 [V, C]");
    TORCH_CHECK(stats_in.dim() == 3 && stats_in.size(2) == C && stats_in.is_contiguous(), "stats_in: [nparts, 2, C]");
    auto rs = [](OptT &t) -> float * { return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr; };
    const ms3d_stream_t st = cur();
    at::Tensor bn0 = at::empty({4, C}, x.options()), bn1 = at::empty({4, C}, x.options());
    float *o0 = bn0.data_ptr<float>(), *o1 = bn1.data_ptr<float>();
    check(ms3d_bn_finalize(stats_in.data_ptr<float>(), (int)stats_in.size(0), (long)V, (int)C, (float)eps0, (float)mom0, fptr(g0),
                           fptr(b0), rs(rm0), rs(rv0), o0, o0 + C, o0 + 2 * C, o0 + 3 * C, st), "ms3d_bn_finalize");
    at::Tensor y1 = at::empty({V, C}, x.options()), st1 = at::empty({nparts, 2, C}, x.options());
    check(ms3d_spconv_layer_forward(x.data_ptr<float>(), nullptr, nbr.data_ptr<int>(), (int)V, 27, (int)C, (int)C, 1, o0 + 2 * C,
                                    o0 + 3 * C, 1, nullptr, nullptr, wf1.data_ptr<float>(), y1.data_ptr<float>(),
                                    st1.data_ptr<float>(), iptr(pl_tile_start), iptr(pl_entries), nullptr, nullptr, st),
          "ms3d_spconv_layer_forward");
    check(ms3d_bn_finalize(st1.data_ptr<float>(), (int)nparts, (long)V, (int)C, (float)eps1, (float)mom1, fptr(g1), fptr(b1),
                           rs(rm1), rs(rv1), o1, o1 + C, o1 + 2 * C, o1 + 3 * C, st), "ms3d_bn_finalize");
    at::Tensor y2 = at::empty({V, C}, x.options());
    OptT st2;
    if (want_stats) st2 = at::empty({nparts, 2, C}, x.options());
    check(ms3d_spconv_layer_forward(y1.data_ptr<float>(), nullptr, nbr.data_ptr<int>(), (int)V, 27, (int)C, (int)C, 1, o1 + 2 * C,
                                    o1 + 3 * C, 1, x.data_ptr<float>(), nullptr, wf2.data_ptr<float>(), y2.data_ptr<float>(),
                                    st2.has_value() ? st2->data_ptr<float>() : nullptr, iptr(pl_tile_start), iptr(pl_entries),
                                    nullptr, nullptr, st),
          "ms3d_spconv_layer_forward");
    return {y1, y2, st2, bn0, bn1};
}

at::Tensor gather_rows(const at::Tensor &x, const at::Tensor &idx)
{
    TORCH_CHECK(x.is_cuda() && x.is_contiguous() && idx.is_contiguous() && idx.scalar_type() == at::kLong &&
                x.scalar_type() == at::kFloat && x.dim() == 2, "gather_rows: f32 [V, C] rows and an int64 index");
    at::Tensor out = at::empty({idx.numel(), x.size(1)}, x.options());
    check(ms3d_gather_rows(x.data_ptr<float>(), (const long long *)idx.data_ptr<int64_t>(), (long)idx.numel(), (int)x.size(1),
                           out.data_ptr<float>(), cur()), "ms3d_gather_rows");
    return out;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "host-side fast path of the minsu3d_amd sparse-voxel engine (pybind over libminsu3d_hip.so's C ABI)";
    m.def("conv_layer_forward", &conv_layer_forward);
    m.def("conv_layer_backward", &conv_layer_backward);
    m.def("bn_finalize", &bn_finalize);
    m.def("gather_rows", &gather_rows);
    m.def("res_block_forward", &res_block_forward);
}
