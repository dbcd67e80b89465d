// torch_plugin.cpp -- the torch.utils.cpp_extension shim between PyTorch tensors and the C ABI of libmontage_hip.so.
//
// Exposes, under the names and argument lists of the reference's two pybind modules,
//     upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain) -> Tensor
//                                                       (montage_gan/torch_utils/ops/upfirdn2d.cpp:16-94, binding :98-101)
//     bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp) -> Tensor
//                                                       (montage_gan/torch_utils/ops/bias_act.cpp:32-90, binding :94-97)
// with the same argument validation (message texts are part of the contract: callers and tests match on them), the same
// output allocation rule (x.options(), x.suggest_memory_format() / empty_like) and the same launch context: the device of
// x is made current and the kernels go onto PyTorch's CURRENT stream for that device.  All device work is done by
// mg_upfirdn2d / mg_bias_act (include/montage_hip.h): no kernel lives in this file, nothing here is compiled for the GPU.
// "Absent" tensors arrive as empty tensors (torch.empty([0]) on any device) and are never dereferenced.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>     // PyTorch-ROCm tensors carry device type "cuda"
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <climits>

#include "../../include/montage_hip.h"

namespace {

int abi_dtype(const at::Tensor& t, const char* what)
{
    switch (t.scalar_type()) {
    case at::kFloat: return MG_F32;
    case at::kHalf: return MG_F16;
    case at::kBFloat16: return MG_BF16;
    case at::kDouble: return MG_F64;
    default: TORCH_CHECK(false, what, ": unsupported dtype ", t.scalar_type());
    }
    return -1;
}

void* current_stream_of(const at::Tensor& t)
{
    return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

void raise_on(int rc)
{
    TORCH_CHECK(rc == 0, mg_last_error());
}

const void* ptr_or_null(const at::Tensor& t)
{
    return t.numel() ? t.data_ptr() : nullptr;
}

// same sizes, and same strides wherever a dimension has more than one element
bool layout_matches(const at::Tensor& a, const at::Tensor& b)
{
    if (a.dim() != b.dim()) return false;
    for (int64_t i = 0; i < a.dim(); i++) {
        if (a.size(i) != b.size(i)) return false;
        if (a.size(i) >= 2 && a.stride(i) != b.stride(i)) return false;
    }
    return true;
}

at::Tensor upfirdn2d(at::Tensor x, at::Tensor f, int64_t upx, int64_t upy, int64_t downx, int64_t downy, int64_t padx0, int64_t padx1,
                     int64_t pady0, int64_t pady1, bool flip, double gain)
{
    TORCH_CHECK(x.is_cuda(), "x must reside on CUDA device");
    TORCH_CHECK(f.device() == x.device(), "f must reside on the same device as x");
    TORCH_CHECK(f.scalar_type() == at::kFloat, "f must be float32");
    TORCH_CHECK(x.numel() <= INT_MAX, "x is too large");
    TORCH_CHECK(f.numel() <= INT_MAX, "f is too large");
    TORCH_CHECK(x.dim() == 4, "x must be rank 4");
    TORCH_CHECK(f.dim() == 2, "f must be rank 2");
    TORCH_CHECK(f.size(0) >= 1 && f.size(1) >= 1, "f must be at least 1x1");
    TORCH_CHECK(upx >= 1 && upy >= 1, "upsampling factor must be at least 1");
    TORCH_CHECK(downx >= 1 && downy >= 1, "downsampling factor must be at least 1");

    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard(device_of(x));
    // C integer division (truncation toward zero), as the reference computes the output size
    const int out_w = (int)(((int)x.size(3) * (int)upx + (int)padx0 + (int)padx1 - (int)f.size(1) + (int)downx) / (int)downx);
    const int out_h = (int)(((int)x.size(2) * (int)upy + (int)pady0 + (int)pady1 - (int)f.size(0) + (int)downy) / (int)downy);
    TORCH_CHECK(out_w >= 1 && out_h >= 1, "output must be at least 1x1");
    at::Tensor y = at::empty({x.size(0), x.size(1), out_h, out_w}, x.options(), x.suggest_memory_format());
    TORCH_CHECK(y.numel() <= INT_MAX, "output is too large");
    if (y.numel() == 0) return y;

    int32_t xs[4], ys[4], fs[2];
    int64_t xst[4], yst[4], fst[2];
    for (int i = 0; i < 4; i++) { xs[i] = (int32_t)x.size(i); xst[i] = x.stride(i); ys[i] = (int32_t)y.size(i); yst[i] = y.stride(i); }
    for (int i = 0; i < 2; i++) { fs[i] = (int32_t)f.size(i); fst[i] = f.stride(i); }
    raise_on(mg_upfirdn2d(x.data_ptr(), f.data_ptr<float>(), y.data_ptr(), abi_dtype(x, "upfirdn2d"), xs, xst, fs, fst, ys, yst,
                          (int)upx, (int)upy, (int)downx, (int)downy, (int)padx0, (int)pady0, flip ? 1 : 0, (float)gain, current_stream_of(x)));
    return y;
}

at::Tensor bias_act(at::Tensor x, at::Tensor b, at::Tensor xref, at::Tensor yref, at::Tensor dy, int64_t grad, int64_t dim, int64_t act,
                    double alpha, double gain, double clamp)
{
    TORCH_CHECK(x.is_cuda(), "x must reside on CUDA device");
    TORCH_CHECK(b.numel() == 0 || (b.dtype() == x.dtype() && b.device() == x.device()), "b must have the same dtype and device as x");
    TORCH_CHECK(xref.numel() == 0 || (xref.sizes() == x.sizes() && xref.dtype() == x.dtype() && xref.device() == x.device()),
                "xref must have the same shape, dtype, and device as x");
    TORCH_CHECK(yref.numel() == 0 || (yref.sizes() == x.sizes() && yref.dtype() == x.dtype() && yref.device() == x.device()),
                "yref must have the same shape, dtype, and device as x");
    TORCH_CHECK(dy.numel() == 0 || (dy.sizes() == x.sizes() && dy.dtype() == x.dtype() && dy.device() == x.device()),
                "dy must have the same dtype and device as x");
    TORCH_CHECK(x.numel() <= INT_MAX, "x is too large");
    TORCH_CHECK(b.dim() == 1, "b must have rank 1");
    TORCH_CHECK(b.numel() == 0 || (dim >= 0 && dim < x.dim()), "dim is out of bounds");
    TORCH_CHECK(b.numel() == 0 || b.numel() == x.size(dim), "b has wrong number of elements");
    TORCH_CHECK(grad >= 0, "grad must be non-negative");

    TORCH_CHECK(x.is_non_overlapping_and_dense(), "x must be non-overlapping and dense");
    TORCH_CHECK(b.is_contiguous(), "b must be contiguous");
    TORCH_CHECK(xref.numel() == 0 || layout_matches(xref, x), "xref must have the same layout as x");
    TORCH_CHECK(yref.numel() == 0 || layout_matches(yref, x), "yref must have the same layout as x");
    TORCH_CHECK(dy.numel() == 0 || layout_matches(dy, x), "dy must have the same layout as x");

    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard(device_of(x));
    at::Tensor y = at::empty_like(x);
    TORCH_CHECK(layout_matches(y, x), "y must have the same layout as x");
    if (x.numel() == 0) return y;
    raise_on(mg_bias_act(x.data_ptr(), ptr_or_null(b), ptr_or_null(xref), ptr_or_null(yref), ptr_or_null(dy), y.data_ptr(),
                         abi_dtype(x, "bias_act"), (int)grad, (int)act, (float)alpha, (float)gain, (float)clamp,
                         x.numel(), (int32_t)b.numel(), b.numel() ? x.stride(dim) : 1, current_stream_of(x)));
    return y;
}

} // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "upfirdn2d_plugin / bias_act_plugin of the reference, backed by libmontage_hip.so (include/montage_hip.h)";
    m.def("upfirdn2d", &upfirdn2d);
    m.def("bias_act", &bias_act);
    m.def("abi_version", []() { return mg_abi_version(); });
}
