// gpfq_torch_ext.cpp -- the thin PyTorch-ROCm C++ extension over the C ABI of include/gpfq.h (level 3):
//
//   torch.ops.gpfq.quantize_layer(W, A, X, step, K, mode, lamb, groups, seed, plan) -> (Q, idx, U, usq_seg)
//   torch.ops.gpfq.quantizer(x, step, K, mode, lamb, uniform) -> q
//
// It owns nothing of the algorithm: argument checks (TORCH_CHECK -> RuntimeError, the convention 8(b) asks for),
// output allocation with at::empty on the inputs' device, the CURRENT HIP stream (PyTorch-ROCm presents HIP devices
// under the "cuda" device type, hence the ...MasqueradingAsCUDA accessor), and one call into libgpfq_hip.so
// (gpfq_quantize_layer_f32 = StepAlgorithm._quantize_layer's native part, step_algorithm.py:194-196, :212-247;
// gpfq_quantizer_f32 = the four quantizers, :7-104).  A cooperative launch that gave up waiting for a peer workgroup
// is redone on GPFQ_PLAN_STREAM_ROWS before anything is returned.  HIP dispatch key only: there is no CPU kernel,
// CPU tensors are refused.
// Built by csrc/Makefile with g++ against the torch headers; linked against libgpfq_hip.so ($ORIGIN rpath).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/gpfq.h"

namespace {

void check_cuda_f32(const at::Tensor& t, const char* name)
{
    TORCH_CHECK(t.is_cuda(), "gpfq: ", name, " must be a tensor on the MI355X (cuda) device; there is no CPU path");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "gpfq: ", name, " must be float32");
}

// (m, D) matrix with unit column stride and a leading dimension >= D, as the C ABI takes it
int64_t leading_dim(const at::Tensor& t, const char* name)
{
    TORCH_CHECK(t.dim() == 2, "gpfq: ", name, " must be 2-D");
    TORCH_CHECK(t.size(1) <= 1 || t.stride(1) == 1, "gpfq: ", name, " must have unit column stride");
    const int64_t ld = t.size(0) > 1 ? t.stride(0) : std::max<int64_t>(t.size(1), 1);
    TORCH_CHECK(ld >= t.size(1), "gpfq: ", name, " rows overlap");
    return std::max<int64_t>(ld, 1);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> quantize_layer(const at::Tensor& W, const at::Tensor& A, const at::Tensor& X,
                                                                           double step, int64_t K, int64_t mode, double lamb,
                                                                           int64_t groups, int64_t seed, int64_t plan)
{
    check_cuda_f32(W, "W");
    check_cuda_f32(A, "A");
    check_cuda_f32(X, "X");
    TORCH_CHECK(W.dim() == 2 && W.is_contiguous(), "gpfq: W must be a contiguous (N, d_g) matrix");
    TORCH_CHECK(groups >= 1 && W.size(0) % groups == 0, "gpfq: out_channels must be divisible by groups");
    const int64_t N = W.size(0), dg = W.size(1), m = A.size(0);
    const int64_t lda = leading_dim(A, "A"), ldx = leading_dim(X, "X");
    TORCH_CHECK(A.size(1) == groups * dg && X.sizes() == A.sizes(), "gpfq: layer inputs must be (m, groups*d_g)");
    TORCH_CHECK(A.device() == W.device() && X.device() == W.device(), "gpfq: all tensors must be on one device");
    TORCH_CHECK(mode >= 0 && mode <= 3, "gpfq: mode must be 0 (msq), 1 (soft), 2 (hard) or 3 (stochastic)");
    TORCH_CHECK(K >= 1 && K <= 32766, "gpfq: boundary index K out of range");
    const bool i16 = K > 126;                          // int8 holds K <= 126 (bits <= 7); bits = 8 needs int16
    c10::DeviceGuard guard(W.device());
    auto opts = W.options();
    at::Tensor Q = at::empty({N, dg}, opts);
    at::Tensor idx = at::empty({N, dg}, opts.dtype(i16 ? at::kShort : at::kChar));
    at::Tensor U = at::empty({N, m}, opts);
    const int64_t mp = gpfq_padded_m(m);
    at::Tensor usq = at::empty({N, mp / 1024}, opts);
    if (N == 0 || dg == 0) {
        Q.zero_(); idx.zero_(); U.zero_(); usq.zero_();
        return {Q, idx, U, usq};
    }
    const size_t wsb = gpfq_workspace_bytes(N, dg, m, (int)groups);
    // only the head of the workspace (exchange granules + status words) must start zeroed: the column buffers and the
    // norms behind it are overwritten in full by the column preparation (hundreds of MB for a ResNet-50 3x3 layer)
    at::Tensor ws = at::empty({(int64_t)wsb}, opts.dtype(at::kByte));
    ws.narrow(0, 0, (int64_t)gpfq_scratch_bytes()).zero_();
    void* stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(W.device().index()).stream();
    auto run = [&](int p) {
        const int rc = gpfq_quantize_layer_f32(W.data_ptr<float>(), A.data_ptr<float>(), lda, X.data_ptr<float>(), ldx, N, dg, m,
                                               (int)groups, (float)step, (int)K, (int)mode, (float)lamb, (uint64_t)seed, 0,
                                               Q.data_ptr<float>(), idx.data_ptr(), i16 ? 2 : 1, U.data_ptr<float>(),
                                               usq.data_ptr<float>(), ws.data_ptr(), wsb, p, stream);
        TORCH_CHECK(rc == 0, "gpfq error ", rc, ": ", gpfq_last_error());
    };
    run((int)plan);
    if (gpfq_last_launch_used_exchange()) {
        int st[4] = {0, 0, 0, 0};
        const int rc = gpfq_read_status(ws.data_ptr(), st, stream);
        TORCH_CHECK(rc == 0 || rc == GPFQ_ERR_TIMEOUT, "gpfq error ", rc, ": ", gpfq_last_error());
        if (rc == GPFQ_ERR_TIMEOUT) run(GPFQ_PLAN_STREAM_ROWS);           // never hand back what a timed-out launch left behind
    }
    return {Q, idx, U, usq};
}

at::Tensor quantizer(const at::Tensor& x, double step, int64_t K, int64_t mode, double lamb, const c10::optional<at::Tensor>& uniform)
{
    check_cuda_f32(x, "x");
    TORCH_CHECK(mode >= 0 && mode <= 3 && K >= 1, "gpfq: bad mode / K");
    c10::DeviceGuard guard(x.device());
    at::Tensor xc = x.contiguous();
    at::Tensor out = at::empty_like(xc);
    const float* un = nullptr;
    at::Tensor uc;
    if (uniform.has_value()) {
        check_cuda_f32(*uniform, "uniform");
        TORCH_CHECK(uniform->numel() == x.numel(), "gpfq: uniform must have one draw per element");
        uc = uniform->contiguous();
        un = uc.data_ptr<float>();
    }
    TORCH_CHECK(mode != 3 || un, "gpfq: the stochastic quantizer needs the uniform draws");
    const int rc = gpfq_quantizer_f32((int)mode, (float)step, xc.data_ptr<float>(), xc.numel(), (int)K, (float)lamb, un,
                                      out.data_ptr<float>(), nullptr, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.device().index()).stream());
    TORCH_CHECK(rc == 0, "gpfq error ", rc, ": ", gpfq_last_error());
    return out.view(x.sizes());
}

}  // namespace

TORCH_LIBRARY(gpfq, m)
{
    m.def("quantize_layer(Tensor W, Tensor A, Tensor X, float step, int K, int mode, float lamb, int groups, int seed, int plan) -> "
          "(Tensor, Tensor, Tensor, Tensor)");
    m.def("quantizer(Tensor x, float step, int K, int mode, float lamb, Tensor? uniform) -> Tensor");
}

TORCH_LIBRARY_IMPL(gpfq, CUDA, m)      // "CUDA" is the HIP dispatch key under PyTorch-ROCm
{
    m.impl("quantize_layer", quantize_layer);
    m.impl("quantizer", quantizer);
}
