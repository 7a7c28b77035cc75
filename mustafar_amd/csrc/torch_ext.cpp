// torch_ext.cpp -- the reference's `mustafar_package` PyTorch extension (kernel/kernel_wrapper/{mustafar_wrapper.cu,
// pybind.cpp}) rebuilt for PyTorch-ROCm on top of the C ABI of include/mustafar_hip.h.  Same module name, same two
// functions, same positional arguments, same checks and exception types (mustafar_wrapper.cu:36-73, :156-194).
// Host code only (no device code in this file): it validates, allocates the output with torch, takes torch's
// current HIP stream and calls libmustafar_hip.so.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPGraphsC10Utils.h>

#include <map>
#include <mutex>
#include <stdexcept>
#include <utility>

#include "../../include/mustafar_hip.h"

namespace {

void common_checks(const torch::Tensor& bmp, const torch::Tensor& NZ, const torch::Tensor& idx, const torch::Tensor& NZ_Offset,
                   const torch::Tensor& B, bool require_B_contiguous)
{
    if (B.device() != bmp.device() || B.device() != NZ.device() || B.device() != idx.device() || B.device() != NZ_Offset.device())
        throw std::runtime_error("All input tensors must be on the same device.");
    if (B.dtype() != at::kHalf) throw std::runtime_error("Tensor B must be of type float16.");
    if (NZ.dtype() != at::kHalf) throw std::runtime_error("Tensor NZ must be of type float16.");
    if (bmp.dtype() != at::kLong) throw std::runtime_error("Tensor bmp must be of type int64.");
    if (idx.dtype() != at::kInt) throw std::runtime_error("Tensor idx must be of type int.");
    if (NZ_Offset.dtype() != at::kInt) throw std::runtime_error("Tensor NZ_Offset must be of type int.");
    TORCH_CHECK(bmp.is_contiguous() && NZ.is_contiguous() && idx.is_contiguous() && NZ_Offset.is_contiguous() &&
                    (!require_B_contiguous || B.is_contiguous()),
                "bmp, NZ, idx, B, C, and Reduction_Workspace tensors must be contiguous.");
    TORCH_CHECK(bmp.is_cuda() && NZ.is_cuda() && idx.is_cuda() && B.is_cuda() && NZ_Offset.is_cuda(),
                "bmp, NZ, idx, B, C, and (not)Reduction_Workspace tensors must be on CUDA device.");
}

int rows_of(const torch::Tensor& B, int Batch_Size, int inner)
{
    TORCH_CHECK(Batch_Size > 0 && inner > 0 && B.numel() % ((int64_t)Batch_Size * inner) == 0, "Tensor B does not hold Batch_Size*N*", inner,
                " elements");
    const int N = (int)(B.numel() / ((int64_t)Batch_Size * inner));
    TORCH_CHECK(N == 1 || N == 8, "Tensor B must hold 1 or 8 rows per batch entry (got ", N, ")");
    return N;
}

void cache_checks(const torch::Tensor& bmp, const torch::Tensor& idx, const torch::Tensor& NZ_Offset, int64_t tiles, int Batch_Size, int groups)
{
    TORCH_CHECK(groups >= 1 && Batch_Size % groups == 0, "Batch_Size must be a multiple of num_key_value_groups");
    const int64_t heads = Batch_Size / groups;
    TORCH_CHECK(bmp.numel() == heads * tiles && idx.numel() == heads * (tiles + 1) && NZ_Offset.numel() == heads,
                "compressed cache does not match the arguments: expected ", heads, " heads x ", tiles, " tiles");
}

}  // namespace

torch::Tensor mustafar_key_formulation(torch::Tensor bmp, torch::Tensor NZ, torch::Tensor idx, torch::Tensor NZ_Offset, torch::Tensor B,
                                       int M_Global, int K_Global, int Batch_Size, int num_key_value_groups)
{
    common_checks(bmp, NZ, idx, NZ_Offset, B, true);
    TORCH_CHECK(K_Global == 128 && M_Global > 0 && M_Global % 64 == 0,
                "mustafar_key_formulation: need K_Global == 128 and M_Global a positive multiple of 64");
    const int N = rows_of(B, Batch_Size, K_Global);
    cache_checks(bmp, idx, NZ_Offset, (int64_t)M_Global * K_Global / 64, Batch_Size, num_key_value_groups);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(B.device());   // PyTorch-ROCm presents HIP devices as "cuda"
    auto C = torch::empty({Batch_Size, N, M_Global}, B.options());   // every element is written by the kernel
    const int err = Key_SplitK_API(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(), nullptr,
                                   reinterpret_cast<const uint64_t*>(bmp.data_ptr<int64_t>()), NZ.data_ptr(),
                                   reinterpret_cast<const uint32_t*>(idx.data_ptr<int32_t>()),
                                   reinterpret_cast<const uint32_t*>(NZ_Offset.data_ptr<int32_t>()), B.data_ptr(), C.data_ptr(),
                                   M_Global, N, K_Global, nullptr, 1, Batch_Size, num_key_value_groups);
    TORCH_CHECK(err == 0, "Key_SplitK_API failed: HIP error ", err);
    return C;
}

torch::Tensor mustafar_value_formulation(torch::Tensor bmp, torch::Tensor NZ, torch::Tensor idx, torch::Tensor NZ_Offset, torch::Tensor B,
                                         torch::Tensor Reduction_Workspace, int M_Global, int K_Global, int Batch_Size,
                                         int num_key_value_groups)
{
    (void)Reduction_Workspace;   // the model's 1-element tensor (llama_mustafar_kernel.py:658); an internal fp32 workspace is used
    common_checks(bmp, NZ, idx, NZ_Offset, B, false);
    TORCH_CHECK(M_Global == 128 && K_Global > 0 && K_Global % 64 == 0,
                "mustafar_value_formulation: need M_Global == 128 and K_Global a positive multiple of 64");
    if (!B.is_contiguous()) B = B.contiguous();
    const int N = rows_of(B, Batch_Size, K_Global);
    cache_checks(bmp, idx, NZ_Offset, (int64_t)M_Global * K_Global / 64, Batch_Size, num_key_value_groups);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(B.device());   // PyTorch-ROCm presents HIP devices as "cuda"
    const int split = mustafar_value_pick_split_k(M_Global, N, K_Global, Batch_Size, num_key_value_groups);
    const int64_t need = mustafar_value_workspace_bytes(M_Global, N, K_Global, Batch_Size, num_key_value_groups, split);
    // fp32 partial slabs: one buffer per (device, stream) -- launches of a stream run in order and may share it; different
    // streams (or threads on different streams) never share slabs
    void* ws = nullptr;
    torch::Tensor captured_ws;   // under a graph capture the slabs come from the capture's own pool (the graph keeps them for its
                                 // replays; a buffer cached here across captures could outlive the pool it came from)
    if (need > 0 && c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None) {
        captured_ws = torch::empty({need}, torch::TensorOptions().dtype(torch::kUInt8).device(B.device()));
        ws = captured_ws.data_ptr();
    } else if (need > 0) {
        auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA();
        static std::mutex mu;
        static std::map<std::pair<int, void*>, torch::Tensor> slabs;
        std::lock_guard<std::mutex> lock(mu);
        auto& t = slabs[{(int)B.device().index(), (void*)stream.stream()}];
        if (!t.defined() || t.numel() < need)
            t = torch::empty({std::max<int64_t>(need, 1 << 20)}, torch::TensorOptions().dtype(torch::kUInt8).device(B.device()));
        ws = t.data_ptr();
    }
    auto C = torch::empty({Batch_Size, N, M_Global}, B.options());
    const int err = Value_SplitK_API(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(), nullptr,
                                     reinterpret_cast<const uint64_t*>(bmp.data_ptr<int64_t>()), NZ.data_ptr(),
                                     reinterpret_cast<const uint32_t*>(idx.data_ptr<int32_t>()),
                                     reinterpret_cast<const uint32_t*>(NZ_Offset.data_ptr<int32_t>()), B.data_ptr(), C.data_ptr(),
                                     M_Global, N, K_Global, ws, split, Batch_Size, num_key_value_groups);
    TORCH_CHECK(err == 0, "Value_SplitK_API failed: HIP error ", err);
    return C;
}

PYBIND11_MODULE(mustafar_package, m)
{
    m.doc() = "PyTorch-ROCm extension for the Mustafar batched SpMV kernels (MI355X / gfx950)";
    m.def("mustafar_key_formulation", &mustafar_key_formulation, "sparse q.K^T over the bitmap-compressed key cache");
    m.def("mustafar_value_formulation", &mustafar_value_formulation, "sparse p.V over the bitmap-compressed value cache");
}
