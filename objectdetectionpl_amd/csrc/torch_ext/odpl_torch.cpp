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

// act(bn(x)); falls back to ATen (returns an undefined tensor) when the layout is not fusable
Tensor bn_act(Tensor x, Tensor weight, Tensor bias, Tensor running_mean, Tensor running_var, Tensor num_batches_tracked,
              bool training, double momentum, double eps, int64_t act, c10::optional<Tensor> residual) {
    if (!fusable(x, weight)) return Tensor();
    c10::optional<Tensor> res;
    if (residual.has_value() && residual->defined()) {
        res = *residual;
        if (res->sizes() != x.sizes() || res->scalar_type() != at::kFloat || !res->is_contiguous(at::MemoryFormat::ChannelsLast)) return Tensor();
    }
    return BnActFn::apply(x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps, act, res);
}
}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("bn_act", &bn_act, "fused BatchNorm2d + activation [+ residual] (channels-last fp32); returns None when not fusable",
          pybind11::arg("x"), pybind11::arg("weight"), pybind11::arg("bias"), pybind11::arg("running_mean"), pybind11::arg("running_var"),
          pybind11::arg("num_batches_tracked"), pybind11::arg("training"), pybind11::arg("momentum"), pybind11::arg("eps"), pybind11::arg("act"),
          pybind11::arg("residual") = c10::optional<Tensor>());
}
