// Thin torch C++ binding for the per-layer epilogue ops of libodpl_hip.so.
// Why C++: a Python autograd.Function + ctypes call costs ~100 us of host time; YOLOv4 has 107 BN layers, forward
// and backward -> the step would be host-bound.  This binding does the same calls with ~10 us of dispatch.
// The detector criteria keep their ctypes host (one call per step: host cost irrelevant).
// Activations may be fp32 or bf16 (a bf16-autocast network); affine parameters and statistics are fp32.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include "../../../include/odpl_hip.h"

namespace {
using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed with code ", rc); }

bool act_dtype(const Tensor& t) { return t.scalar_type() == at::kFloat || t.scalar_type() == at::kBFloat16; }

bool fusable(const Tensor& x, const Tensor& w) {
    return x.is_cuda() && act_dtype(x) && x.dim() == 4 && x.size(1) % 4 == 0 && w.defined() && w.scalar_type() == at::kFloat &&
           x.is_contiguous(at::MemoryFormat::ChannelsLast);
}

hipStream_t stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// element offset -> byte pointer inside a [R, CT] activation matrix (column slice of a concatenated tensor)
void* at_channel(const Tensor& t, int64_t channel) { return static_cast<char*>(t.data_ptr()) + channel * t.element_size(); }

struct Stats { Tensor mean, invstd; };

// one forward launch group: y (row stride y_rs, column offset already applied) = act(bn(x)) [+ residual]
Stats launch_forward(const Tensor& x, const Tensor& residual, const Tensor& weight, const Tensor& bias, const Tensor& running_mean,
                     const Tensor& running_var, const Tensor& nbt, bool training, double momentum, double eps, int64_t act, void* y, int64_t y_rs,
                     const Stats* stats_out = nullptr) {
    const int64_t C = x.size(1), R = x.numel() / C;
    Stats s;
    if (training && stats_out != nullptr && stats_out->mean.defined()) {
        s = *stats_out;
    } else if (training) {
        s.mean = at::empty({C}, weight.options());
        s.invstd = at::empty({C}, weight.options());
    } else {
        s.mean = running_mean.clone();
        s.invstd = at::rsqrt(running_var + eps);
    }
    const size_t need = odpl_bn_act_workspace_bytes(R, (int32_t)C);
    auto ws = at::empty({(int64_t)need}, x.options().dtype(at::kByte));
    check(odpl_bn_act_forward_strided(x.data_ptr(), residual.defined() ? residual.data_ptr() : nullptr, x.scalar_type() == at::kBFloat16 ? 1 : 0,
                                      R, (int32_t)C, weight.data_ptr<float>(), bias.data_ptr<float>(), running_mean.data_ptr<float>(),
                                      running_var.data_ptr<float>(), training ? 1 : 0, (float)momentum, (float)eps, (int32_t)act,
                                      s.mean.data_ptr<float>(), s.invstd.data_ptr<float>(), y, y_rs, ws.data_ptr(), need,
                                      (training && nbt.defined()) ? nbt.data_ptr<int64_t>() : nullptr, stream_of(x)),
          "odpl_bn_act_forward_strided");
    return s;
}

struct Grads { Tensor gx, dgamma, dbeta; };

Grads launch_backward(const Tensor& x, const void* gy, int64_t gy_rs, const Tensor& weight, const Tensor& bias, const Tensor& mean,
                      const Tensor& invstd, bool training, int64_t act, const Tensor& dgamma_out = Tensor(), const Tensor& dbeta_out = Tensor()) {
    const int64_t C = x.size(1), R = x.numel() / C;
    Grads g{at::empty_like(x), dgamma_out.defined() ? dgamma_out : at::empty({C}, weight.options()),
            dbeta_out.defined() ? dbeta_out : at::empty({C}, weight.options())};
    const size_t need = odpl_bn_act_workspace_bytes(R, (int32_t)C);
    auto ws = at::empty({(int64_t)need}, x.options().dtype(at::kByte));
    check(odpl_bn_act_backward_strided(x.data_ptr(), gy, gy_rs, x.scalar_type() == at::kBFloat16 ? 1 : 0, R, (int32_t)C, weight.data_ptr<float>(),
                                       bias.data_ptr<float>(), mean.data_ptr<float>(), invstd.data_ptr<float>(), training ? 1 : 0, (int32_t)act,
                                       g.gx.data_ptr(), g.dgamma.data_ptr<float>(), g.dbeta.data_ptr<float>(), ws.data_ptr(), need, stream_of(x)),
          "odpl_bn_act_backward_strided");
    return g;
}

Tensor as_grad_layout(const Tensor& gy, const Tensor& like) {
    if (gy.is_contiguous(at::MemoryFormat::ChannelsLast) && gy.scalar_type() == like.scalar_type()) return gy;
    return gy.to(like.scalar_type()).contiguous(at::MemoryFormat::ChannelsLast);
}

struct BnActFn : public torch::autograd::Function<BnActFn> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor weight, Tensor bias, Tensor running_mean, Tensor running_var,
                          Tensor nbt, bool training, double momentum, double eps, int64_t act, c10::optional<Tensor> residual_opt) {
        const Tensor residual = residual_opt.has_value() ? *residual_opt : Tensor();
        auto y = at::empty_like(x);
        const Stats s = launch_forward(x, residual, weight, bias, running_mean, running_var, nbt, training, momentum, eps, act, y.data_ptr(), x.size(1));
        ctx->save_for_backward({x, weight, bias, s.mean, s.invstd});
        ctx->saved_data["training"] = training;
        ctx->saved_data["act"] = act;
        ctx->saved_data["has_res"] = residual.defined();
        return y;
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
        auto sv = ctx->get_saved_variables();
        const Tensor gy = as_grad_layout(grads[0], sv[0]);
        const Grads g = launch_backward(sv[0], gy.data_ptr(), sv[0].size(1), sv[1], sv[2], sv[3], sv[4], ctx->saved_data["training"].toBool(),
                                        ctx->saved_data["act"].toInt());
        return {g.gx, g.dgamma, g.dbeta, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                ctx->saved_data["has_res"].toBool() ? grads[0] : Tensor()};
    }
};

// cat((act1(bn1(x1)), act2(bn2(x2))), 1) without the cat: both epilogues write their column slice of one [R, C1+C2] buffer,
// backward reads the two slices of the incoming gradient in place (the CSP merges, utils/YoloV4Utils.py:264-398).
struct BnActCatFn : public torch::autograd::Function<BnActCatFn> {
    static Tensor forward(AutogradContext* ctx, Tensor x1, Tensor w1, Tensor b1, Tensor rm1, Tensor rv1, Tensor nbt1,
                          Tensor x2, Tensor w2, Tensor b2, Tensor rm2, Tensor rv2, Tensor nbt2,
                          bool training, double mom1, double eps1, int64_t act1, double mom2, double eps2, int64_t act2) {
        const int64_t C1 = x1.size(1), CT = C1 + x2.size(1);
        auto y = at::empty({x1.size(0), CT, x1.size(2), x1.size(3)}, x1.options().memory_format(at::MemoryFormat::ChannelsLast));
        const Stats s1 = launch_forward(x1, Tensor(), w1, b1, rm1, rv1, nbt1, training, mom1, eps1, act1, at_channel(y, 0), CT);
        const Stats s2 = launch_forward(x2, Tensor(), w2, b2, rm2, rv2, nbt2, training, mom2, eps2, act2, at_channel(y, C1), CT);
        ctx->save_for_backward({x1, w1, b1, s1.mean, s1.invstd, x2, w2, b2, s2.mean, s2.invstd});
        ctx->saved_data["training"] = training;
        ctx->saved_data["act1"] = act1;
        ctx->saved_data["act2"] = act2;
        return y;
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
        auto sv = ctx->get_saved_variables();
        const bool training = ctx->saved_data["training"].toBool();
        const Tensor gy = as_grad_layout(grads[0], sv[0]);
        const int64_t C1 = sv[0].size(1), CT = gy.size(1);
        const Grads g1 = launch_backward(sv[0], at_channel(gy, 0), CT, sv[1], sv[2], sv[3], sv[4], training, ctx->saved_data["act1"].toInt());
        const Grads g2 = launch_backward(sv[5], at_channel(gy, C1), CT, sv[6], sv[7], sv[8], sv[9], training, ctx->saved_data["act2"].toInt());
        return {g1.gx, g1.dgamma, g1.dbeta, Tensor(), Tensor(), Tensor(), g2.gx, g2.dgamma, g2.dbeta, Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor bn_act_cat(Tensor x1, Tensor w1, Tensor b1, Tensor rm1, Tensor rv1, Tensor nbt1, double mom1, double eps1, int64_t act1,
                  Tensor x2, Tensor w2, Tensor b2, Tensor rm2, Tensor rv2, Tensor nbt2, double mom2, double eps2, int64_t act2, bool training) {
    if (!fusable(x1, w1) || !fusable(x2, w2) || x1.scalar_type() != x2.scalar_type() || x1.size(0) != x2.size(0) || x1.size(2) != x2.size(2) ||
        x1.size(3) != x2.size(3))
        return Tensor();
    return BnActCatFn::apply(x1, w1, b1, rm1, rv1, nbt1, x2, w2, b2, rm2, rv2, nbt2, training, mom1, eps1, act1, mom2, eps2, act2);
}

// act(bn(cat((x1, x2), 1))) for ONE BatchNorm over the concatenated channels (YOLOv5 BottleneckCSP, utils/YoloV5Utils.py:76-93):
// per-channel statistics make it two independent epilogues over the halves of the module's parameters and buffers.  The halves
// are narrow() views taken HERE (inside the Function: no autograd nodes, no slice_backward launches), the parameter gradients are
// written straight into the halves of one [C1+C2] buffer.
struct BnOfCatFn : public torch::autograd::Function<BnOfCatFn> {
    static Tensor forward(AutogradContext* ctx, Tensor x1, Tensor x2, Tensor w, Tensor b, Tensor rm, Tensor rv, Tensor nbt,
                          bool training, double mom, double eps, int64_t act) {
        const int64_t C1 = x1.size(1), C2 = x2.size(1), CT = C1 + C2;
        auto y = at::empty({x1.size(0), CT, x1.size(2), x1.size(3)}, x1.options().memory_format(at::MemoryFormat::ChannelsLast));
        Stats full;
        if (training) { full.mean = at::empty({CT}, w.options()); full.invstd = at::empty({CT}, w.options()); }
        else { full.mean = rm.clone(); full.invstd = at::rsqrt(rv + eps); }
        const Stats h1{full.mean.narrow(0, 0, C1), full.invstd.narrow(0, 0, C1)}, h2{full.mean.narrow(0, C1, C2), full.invstd.narrow(0, C1, C2)};
        if (training) {
            launch_forward(x1, Tensor(), w.narrow(0, 0, C1), b.narrow(0, 0, C1), rm.narrow(0, 0, C1), rv.narrow(0, 0, C1), nbt, true, mom, eps, act,
                           at_channel(y, 0), CT, &h1);
            launch_forward(x2, Tensor(), w.narrow(0, C1, C2), b.narrow(0, C1, C2), rm.narrow(0, C1, C2), rv.narrow(0, C1, C2), Tensor(), true, mom, eps, act,
                           at_channel(y, C1), CT, &h2);                            // num_batches_tracked counts the module once
        } else {
            const Tensor halves[2][5] = {{x1, w.narrow(0, 0, C1), b.narrow(0, 0, C1), h1.mean, h1.invstd}, {x2, w.narrow(0, C1, C2), b.narrow(0, C1, C2), h2.mean, h2.invstd}};
            for (int k = 0; k < 2; ++k) {
                const Tensor& x = halves[k][0];
                const int64_t C = x.size(1), R = x.numel() / C;
                check(odpl_bn_act_forward_strided(x.data_ptr(), nullptr, x.scalar_type() == at::kBFloat16 ? 1 : 0, R, (int32_t)C,
                                                  halves[k][1].data_ptr<float>(), halves[k][2].data_ptr<float>(), nullptr, nullptr, 0, 0.f, (float)eps, (int32_t)act,
                                                  halves[k][3].data_ptr<float>(), halves[k][4].data_ptr<float>(), at_channel(y, k == 0 ? 0 : C1), CT, nullptr, 0, nullptr,
                                                  stream_of(x)), "odpl_bn_act_forward_strided");
            }
        }
        ctx->save_for_backward({x1, x2, w, b, full.mean, full.invstd});
        ctx->saved_data["training"] = training;
        ctx->saved_data["act"] = act;
        return y;
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
        auto sv = ctx->get_saved_variables();
        const bool training = ctx->saved_data["training"].toBool();
        const int64_t act = ctx->saved_data["act"].toInt();
        const Tensor &x1 = sv[0], &x2 = sv[1], &w = sv[2], &b = sv[3], &mean = sv[4], &invstd = sv[5];
        const Tensor gy = as_grad_layout(grads[0], x1);
        const int64_t C1 = x1.size(1), C2 = x2.size(1), CT = C1 + C2;
        auto dg = at::empty({CT}, w.options()), db = at::empty({CT}, w.options());
        const Grads g1 = launch_backward(x1, at_channel(gy, 0), CT, w.narrow(0, 0, C1), b.narrow(0, 0, C1), mean.narrow(0, 0, C1), invstd.narrow(0, 0, C1),
                                         training, act, dg.narrow(0, 0, C1), db.narrow(0, 0, C1));
        const Grads g2 = launch_backward(x2, at_channel(gy, C1), CT, w.narrow(0, C1, C2), b.narrow(0, C1, C2), mean.narrow(0, C1, C2), invstd.narrow(0, C1, C2),
                                         training, act, dg.narrow(0, C1, C2), db.narrow(0, C1, C2));
        return {g1.gx, g2.gx, dg, db, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor bn_act_of_cat(Tensor x1, Tensor x2, Tensor w, Tensor b, Tensor rm, Tensor rv, Tensor nbt, bool training, double mom, double eps, int64_t act) {
    if (!fusable(x1, w) || !fusable(x2, w) || x1.scalar_type() != x2.scalar_type() || x1.size(0) != x2.size(0) || x1.size(2) != x2.size(2) ||
        x1.size(3) != x2.size(3) || w.numel() != x1.size(1) + x2.size(1))
        return Tensor();
    return BnOfCatFn::apply(x1, x2, w, b, rm, rv, nbt, training, mom, eps, act);
}

// inference (eval mode, no gradient wanted): one launch, statistics read from the running buffers inside the kernel
Tensor bn_act_infer(const Tensor& x, const Tensor& weight, const Tensor& bias, const Tensor& running_mean, const Tensor& running_var,
                    double eps, int64_t act, const Tensor& residual) {
    const int64_t C = x.size(1), R = x.numel() / C;
    auto y = at::empty_like(x);
    check(odpl_bn_act_forward_strided(x.data_ptr(), residual.defined() ? residual.data_ptr() : nullptr, x.scalar_type() == at::kBFloat16 ? 1 : 0,
                                      R, (int32_t)C, weight.data_ptr<float>(), bias.data_ptr<float>(), running_mean.data_ptr<float>(),
                                      running_var.data_ptr<float>(), 0, 0.f, (float)eps, (int32_t)act, nullptr, nullptr, y.data_ptr(), C, nullptr, 0,
                                      nullptr, stream_of(x)),
          "odpl_bn_act_forward_strided");
    return y;
}

// act(bn(x)) [+ residual]; returns an undefined tensor (None) when the layout is not fusable
Tensor bn_act(Tensor x, Tensor weight, Tensor bias, Tensor running_mean, Tensor running_var, Tensor num_batches_tracked,
              bool training, double momentum, double eps, int64_t act, c10::optional<Tensor> residual) {
    if (!fusable(x, weight)) return Tensor();
    c10::optional<Tensor> res;
    if (residual.has_value() && residual->defined()) {
        res = *residual;
        if (res->sizes() != x.sizes() || res->scalar_type() != x.scalar_type() || !res->is_contiguous(at::MemoryFormat::ChannelsLast)) return Tensor();
    }
    const bool wants_grad = at::GradMode::is_enabled() && (x.requires_grad() || weight.requires_grad() || bias.requires_grad() ||
                                                           (res.has_value() && res->requires_grad()));
    if (!training && !wants_grad) return bn_act_infer(x, weight, bias, running_mean, running_var, eps, act, res.has_value() ? *res : Tensor());
    return BnActFn::apply(x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps, act, res);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("bn_act_of_cat", &bn_act_of_cat, "act(bn(cat((x1, x2), 1))) for one BatchNorm over the concatenated channels, without the cat; None when not fusable");
    m.def("bn_act_cat", &bn_act_cat, "cat((act1(bn1(x1)), act2(bn2(x2))), 1) written in place by the two epilogues; None when not fusable");
    m.def("bn_act", &bn_act, "fused BatchNorm2d + activation [+ residual] (channels-last fp32 / bf16); returns None when not fusable",
          pybind11::arg("x"), pybind11::arg("weight"), pybind11::arg("bias"), pybind11::arg("running_mean"), pybind11::arg("running_var"),
          pybind11::arg("num_batches_tracked"), pybind11::arg("training"), pybind11::arg("momentum"), pybind11::arg("eps"), pybind11::arg("act"),
          pybind11::arg("residual") = c10::optional<Tensor>());
}
