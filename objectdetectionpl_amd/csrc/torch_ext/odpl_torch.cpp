// Thin torch C++ binding for the per-layer epilogue ops of libodpl_hip.so.
// Why C++: a Python autograd.Function + ctypes call costs ~100 us of host time; YOLOv4 has 107 BN layers, forward
// and backward -> the step would be host-bound.  This binding does the same calls with ~10 us of dispatch.
// The detector criteria keep their ctypes host (one call per step: host cost irrelevant).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include "../../../include/odpl_hip.h"

namespace {
using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed with code ", rc); }

bool fusable(const Tensor& x, const Tensor& w) {
    return x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 4 && x.size(1) % 4 == 0 && w.defined() &&
           x.is_contiguous(at::MemoryFormat::ChannelsLast);
}

struct BnActFn : public torch::autograd::Function<BnActFn> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor weight, Tensor bias, Tensor running_mean, Tensor running_var,
                          Tensor nbt, bool training, double momentum, double eps, int64_t act, c10::optional<Tensor> residual_opt) {
        const Tensor residual = residual_opt.has_value() ? *residual_opt : Tensor();
        const int64_t C = x.size(1), R = x.numel() / C;
        auto y = at::empty_like(x);
        Tensor mean, invstd;
        if (training) {
            mean = at::empty({C}, x.options());
            invstd = at::empty({C}, x.options());
        } else {
            mean = running_mean.clone();
            invstd = at::rsqrt(running_var + eps);
        }
        const size_t need = odpl_bn_act_workspace_bytes(R, (int32_t)C);
        auto ws = at::empty({(int64_t)need}, x.options().dtype(at::kByte));
        auto stream = c10::hip::getCurrentHIPStream(x.device().index()).stream();
        check(odpl_bn_act_forward_residual(x.data_ptr<float>(), residual.defined() ? residual.data_ptr<float>() : nullptr, R, (int32_t)C,
                                  weight.data_ptr<float>(), bias.data_ptr<float>(),
                                  running_mean.data_ptr<float>(), running_var.data_ptr<float>(), training ? 1 : 0, (float)momentum,
                                  (float)eps, (int32_t)act, mean.data_ptr<float>(), invstd.data_ptr<float>(), y.data_ptr<float>(),
                                  ws.data_ptr(), need, (training && nbt.defined()) ? nbt.data_ptr<int64_t>() : nullptr, stream),
              "odpl_bn_act_forward");
        ctx->save_for_backward({x, weight, bias, mean, invstd});
        ctx->saved_data["training"] = training;
        ctx->saved_data["act"] = act;
        ctx->saved_data["has_res"] = residual.defined();
        return y;
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
        auto saved = ctx->get_saved_variables();
        const Tensor &x = saved[0], &weight = saved[1], &bias = saved[2], &mean = saved[3], &invstd = saved[4];
        const bool training = ctx->saved_data["training"].toBool();
        const int64_t act = ctx->saved_data["act"].toInt();
        const int64_t C = x.size(1), R = x.numel() / C;
        Tensor gy = grads[0];
        if (!gy.is_contiguous(at::MemoryFormat::ChannelsLast) || gy.scalar_type() != at::kFloat)
            gy = gy.to(at::kFloat).contiguous(at::MemoryFormat::ChannelsLast);
        auto gx = at::empty_like(x);
        auto dg = at::empty({C}, x.options());
        auto db = at::empty({C}, x.options());
        const size_t need = odpl_bn_act_workspace_bytes(R, (int32_t)C);
        auto ws = at::empty({(int64_t)need}, x.options().dtype(at::kByte));
        auto stream = c10::hip::getCurrentHIPStream(x.device().index()).stream();
        check(odpl_bn_act_backward(x.data_ptr<float>(), gy.data_ptr<float>(), R, (int32_t)C, weight.data_ptr<float>(), bias.data_ptr<float>(),
                                   mean.data_ptr<float>(), invstd.data_ptr<float>(), training ? 1 : 0, (int32_t)act, gx.data_ptr<float>(),
                                   dg.data_ptr<float>(), db.data_ptr<float>(), ws.data_ptr(), need, stream), "odpl_bn_act_backward");
        return {gx, dg, db, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                ctx->saved_data["has_res"].toBool() ? grads[0] : Tensor()};
    }
};

// cat((act1(bn1(x1)), act2(bn2(x2))), 1) without the cat: both epilogues write their column slice of one [R, C1+C2] buffer,
// backward reads the two slices of the incoming gradient in place (the CSP merges, utils/YoloV4Utils.py:264-398).
struct BnActCatFn : public torch::autograd::Function<BnActCatFn> {
    static Tensor forward(AutogradContext* ctx, Tensor x1, Tensor w1, Tensor b1, Tensor rm1, Tensor rv1, Tensor nbt1,
                          Tensor x2, Tensor w2, Tensor b2, Tensor rm2, Tensor rv2, Tensor nbt2,
                          bool training, double mom1, double eps1, int64_t act1, double mom2, double eps2, int64_t act2) {
        const int64_t C1 = x1.size(1), C2 = x2.size(1), R = x1.numel() / C1, CT = C1 + C2;
        auto y = at::empty({x1.size(0), CT, x1.size(2), x1.size(3)}, x1.options().memory_format(at::MemoryFormat::ChannelsLast));
        auto stream = c10::hip::getCurrentHIPStream(x1.device().index()).stream();
        const Tensor* xs[2] = {&x1, &x2}; const Tensor* ws_[2] = {&w1, &w2}; const Tensor* bs[2] = {&b1, &b2};
        const Tensor* rms[2] = {&rm1, &rm2}; const Tensor* rvs[2] = {&rv1, &rv2}; const Tensor* nbts[2] = {&nbt1, &nbt2};
        const double moms[2] = {mom1, mom2}, epss[2] = {eps1, eps2}; const int64_t acts[2] = {act1, act2}, offs[2] = {0, C1};
        Tensor means[2], invstds[2];
        for (int k = 0; k < 2; ++k) {
            const int64_t C = xs[k]->size(1);
            if (training) { means[k] = at::empty({C}, x1.options()); invstds[k] = at::empty({C}, x1.options()); }
            else { means[k] = rms[k]->clone(); invstds[k] = at::rsqrt(*rvs[k] + epss[k]); }
            const size_t need = odpl_bn_act_workspace_bytes(R, (int32_t)C);
            auto ws = at::empty({(int64_t)need}, x1.options().dtype(at::kByte));
            check(odpl_bn_act_forward_strided(xs[k]->data_ptr<float>(), nullptr, R, (int32_t)C, ws_[k]->data_ptr<float>(), bs[k]->data_ptr<float>(),
                                              rms[k]->data_ptr<float>(), rvs[k]->data_ptr<float>(), training ? 1 : 0, (float)moms[k], (float)epss[k],
                                              (int32_t)acts[k], means[k].data_ptr<float>(), invstds[k].data_ptr<float>(),
                                              y.data_ptr<float>() + offs[k], CT, ws.data_ptr(), need,
                                              (training && nbts[k]->defined()) ? nbts[k]->data_ptr<int64_t>() : nullptr, stream),
                  "odpl_bn_act_forward_strided");
        }
        ctx->save_for_backward({x1, w1, b1, means[0], invstds[0], x2, w2, b2, means[1], invstds[1]});
        ctx->saved_data["training"] = training;
        ctx->saved_data["act1"] = act1;
        ctx->saved_data["act2"] = act2;
        return y;
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
        auto sv = ctx->get_saved_variables();
        const bool training = ctx->saved_data["training"].toBool();
        const int64_t acts[2] = {ctx->saved_data["act1"].toInt(), ctx->saved_data["act2"].toInt()};
        Tensor gy = grads[0];
        if (!gy.is_contiguous(at::MemoryFormat::ChannelsLast) || gy.scalar_type() != at::kFloat)
            gy = gy.to(at::kFloat).contiguous(at::MemoryFormat::ChannelsLast);
        const int64_t C1 = sv[0].size(1), CT = gy.size(1), R = sv[0].numel() / C1;
        auto stream = c10::hip::getCurrentHIPStream(gy.device().index()).stream();
        Tensor out[2][3];
        for (int k = 0; k < 2; ++k) {
            const Tensor &x = sv[5 * k], &w = sv[5 * k + 1], &b = sv[5 * k + 2], &mean = sv[5 * k + 3], &invstd = sv[5 * k + 4];
            const int64_t C = x.size(1);
            out[k][0] = at::empty_like(x); out[k][1] = at::empty({C}, x.options()); out[k][2] = at::empty({C}, x.options());
            const size_t need = odpl_bn_act_workspace_bytes(R, (int32_t)C);
            auto ws = at::empty({(int64_t)need}, x.options().dtype(at::kByte));
            check(odpl_bn_act_backward_strided(x.data_ptr<float>(), gy.data_ptr<float>() + (k ? C1 : 0), CT, R, (int32_t)C, w.data_ptr<float>(),
                                               b.data_ptr<float>(), mean.data_ptr<float>(), invstd.data_ptr<float>(), training ? 1 : 0,
                                               (int32_t)acts[k], out[k][0].data_ptr<float>(), out[k][1].data_ptr<float>(),
                                               out[k][2].data_ptr<float>(), ws.data_ptr(), need, stream), "odpl_bn_act_backward_strided");
        }
        return {out[0][0], out[0][1], out[0][2], Tensor(), Tensor(), Tensor(), out[1][0], out[1][1], out[1][2], Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor bn_act_cat(Tensor x1, Tensor w1, Tensor b1, Tensor rm1, Tensor rv1, Tensor nbt1, double mom1, double eps1, int64_t act1,
                  Tensor x2, Tensor w2, Tensor b2, Tensor rm2, Tensor rv2, Tensor nbt2, double mom2, double eps2, int64_t act2, bool training) {
    if (!fusable(x1, w1) || !fusable(x2, w2) || x1.size(0) != x2.size(0) || x1.size(2) != x2.size(2) || x1.size(3) != x2.size(3)) return Tensor();
    return BnActCatFn::apply(x1, w1, b1, rm1, rv1, nbt1, x2, w2, b2, rm2, rv2, nbt2, training, mom1, eps1, act1, mom2, eps2, act2);
}

// inference (eval mode, no gradient wanted): one launch, statistics read from the running buffers inside the kernel
Tensor bn_act_infer(const Tensor& x, const Tensor& weight, const Tensor& bias, const Tensor& running_mean, const Tensor& running_var,
                    double eps, int64_t act, const Tensor& residual) {
    const int64_t C = x.size(1), R = x.numel() / C;
    auto y = at::empty_like(x);
    auto stream = c10::hip::getCurrentHIPStream(x.device().index()).stream();
    check(odpl_bn_act_forward_strided(x.data_ptr<float>(), residual.defined() ? residual.data_ptr<float>() : nullptr, R, (int32_t)C,
                                      weight.data_ptr<float>(), bias.data_ptr<float>(), running_mean.data_ptr<float>(), running_var.data_ptr<float>(),
                                      0, 0.f, (float)eps, (int32_t)act, nullptr, nullptr, y.data_ptr<float>(), C, nullptr, 0, nullptr, stream),
          "odpl_bn_act_forward_strided");
    return y;
}

// act(bn(x)); falls back to ATen (returns an undefined tensor) when the layout is not fusable
Tensor bn_act(Tensor x, Tensor weight, Tensor bias, Tensor running_mean, Tensor running_var, Tensor num_batches_tracked,
              bool training, double momentum, double eps, int64_t act, c10::optional<Tensor> residual) {
    if (!fusable(x, weight)) return Tensor();
    c10::optional<Tensor> res;
    if (residual.has_value() && residual->defined()) {
        res = *residual;
        if (res->sizes() != x.sizes() || res->scalar_type() != at::kFloat || !res->is_contiguous(at::MemoryFormat::ChannelsLast)) return Tensor();
    }
    const bool wants_grad = at::GradMode::is_enabled() && (x.requires_grad() || weight.requires_grad() || bias.requires_grad() ||
                                                           (res.has_value() && res->requires_grad()));
    if (!training && !wants_grad) return bn_act_infer(x, weight, bias, running_mean, running_var, eps, act, res.has_value() ? *res : Tensor());
    return BnActFn::apply(x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps, act, res);
}
}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("bn_act_cat", &bn_act_cat, "cat((act1(bn1(x1)), act2(bn2(x2))), 1) written in place by the two epilogues; None when not fusable");
    m.def("bn_act", &bn_act, "fused BatchNorm2d + activation [+ residual] (channels-last fp32); returns None when not fusable",
          pybind11::arg("x"), pybind11::arg("weight"), pybind11::arg("bias"), pybind11::arg("running_mean"), pybind11::arg("running_var"),
          pybind11::arg("num_batches_tracked"), pybind11::arg("training"), pybind11::arg("momentum"), pybind11::arg("eps"), pybind11::arg("act"),
          pybind11::arg("residual") = c10::optional<Tensor>());
}
