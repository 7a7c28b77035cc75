"""GPU: the reference's own module names (`import mustafar_package`, `import kernel.compression`, model :14/:19) resolved
by mustafar_amd/dropin -- the compiled PyTorch extension over the C ABI -- driven with the reference's exact call sequence
(model :273-275, :313-315).  Results must equal the ctypes mirror bit for bit (same kernels) and the oracle within fp16."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc
from tests.util import fp16_bound

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DROPIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mustafar_amd", "dropin")


@pytest.fixture(scope="module")
def ref_modules():
    sys.path.insert(0, DROPIN)
    try:
        import kernel.compression as compression
        import mustafar_package
        yield mustafar_package, compression
    finally:
        sys.path.remove(DROPIN)


def test_reference_call_sequence_through_compiled_extension(ref_modules):
    mustafar_package, compression = ref_modules
    assert mustafar_package.__file__.endswith(".so"), "the compiled extension must be the one imported"
    from mustafar_amd import mustafar_package as mirror
    torch.manual_seed(0)
    bsz, hq, hkv, T, D = 2, 8, 2, 512, 128
    groups, BH, Bkv = hq // hkv, bsz * hq, bsz * hkv
    K = torch.randn(bsz, hkv, T, D, device=DEV).half()
    Kp = compression.prune_magnitude(K, 0.7)
    k_bmps, k_idxs, k_nzs = compression.convert_key_batched(Kp.reshape(Bkv, -1, D))                 # model :422
    k_nz_offset = torch.zeros(Bkv, dtype=torch.int32, device=DEV)
    for i in range(1, Bkv):                                                                          # model :424-425
        k_nz_offset[i] = k_nz_offset[i - 1] + k_idxs[i - 1][-1] // 4
    q = torch.randn(bsz, hq, 1, D, device=DEV).half()
    padded_query = F.pad(q.view(BH, -1, D), (0, 0, 0, 7), mode="constant", value=0)                 # model :273
    att = mustafar_package.mustafar_key_formulation(k_bmps, torch.cat(k_nzs), k_idxs, k_nz_offset, padded_query, T, D, BH, groups)  # :274
    assert att.shape == (BH, 8, T) and not att[:, 1:].any()
    same = mirror.mustafar_key_formulation(k_bmps, torch.cat(k_nzs), k_idxs, k_nz_offset, padded_query, T, D, BH, groups)
    assert torch.equal(att, same)
    _, Cd = orc.key_spmv(k_bmps.cpu().numpy(), torch.cat(k_nzs).cpu().numpy(), k_idxs.cpu().numpy(), k_nz_offset.cpu().numpy(),
                         padded_query.cpu().numpy(), T, D, BH, groups)
    Kd = Kp.reshape(Bkv, T, D).cpu().numpy().astype(np.float64)
    qn = padded_query.cpu().numpy().astype(np.float64)
    sumabs = np.stack([np.abs(Kd[b // groups]) @ np.abs(qn[b]).T for b in range(BH)]).transpose(0, 2, 1)
    assert (np.abs(att.float().cpu().numpy() - Cd) <= fp16_bound(Cd, sumabs)).all()

    att_c = att[:, 0:1, :].view(bsz, hq, 1, T)                                                       # model :275
    p = torch.softmax(att_c / np.sqrt(D), dim=-1, dtype=torch.float32).half()
    V = torch.randn(bsz, hkv, T, D, device=DEV).half()
    Vp = compression.prune_magnitude(V, 0.7)
    v_bmps, v_idxs, v_nzs = compression.convert_value_batched(Vp.reshape(Bkv, -1, D))
    v_nz_offset = torch.zeros(Bkv, dtype=torch.int32, device=DEV)
    for i in range(1, Bkv):
        v_nz_offset[i] = v_nz_offset[i - 1] + v_idxs[i - 1][-1] // 4
    padded_score = F.pad(p.view(BH, -1, T), (0, 0, 0, 7)).contiguous()                               # model :313
    ws = torch.zeros(1, dtype=torch.float16, device=DEV)                                             # model :658
    out = mustafar_package.mustafar_value_formulation(v_bmps, torch.cat(v_nzs), v_idxs, v_nz_offset, padded_score, ws, D, T, BH, groups)  # :314
    assert out.shape == (BH, 8, D) and not out[:, 1:].any()
    want = torch.matmul(p.float(), Vp.float().repeat_interleave(groups, 1)).view(BH, 1, D)
    torch.testing.assert_close(out[:, 0:1].float(), want, rtol=3e-3, atol=2e-4)
    # error behaviour of the extension = the reference's (mustafar_wrapper.cu:36-73)
    with pytest.raises(RuntimeError, match="float16"):
        mustafar_package.mustafar_key_formulation(k_bmps, torch.cat(k_nzs), k_idxs, k_nz_offset, padded_query.float(), T, D, BH, groups)
    with pytest.raises(RuntimeError, match="same device"):
        mustafar_package.mustafar_key_formulation(k_bmps.cpu(), torch.cat(k_nzs), k_idxs, k_nz_offset, padded_query, T, D, BH, groups)
    with pytest.raises(RuntimeError, match="contiguous"):
        mustafar_package.mustafar_key_formulation(k_bmps, torch.cat(k_nzs), k_idxs, k_nz_offset, padded_query.transpose(1, 2), T, D, BH, groups)


def test_the_hooks_stream_concatenation_costs_nothing(ref_modules):
    """The unchanged hook re-concatenates every head's stream on every decode step (model :274, :314) and rebuilds the list per head at
    every trigger (:368, :390).  With the tensors `convert_*_batched` returns, the first is the packed buffer itself (same storage,
    no kernel) and the second fills one new buffer -- checked here with the model's own statements, on the GPU, K and V."""
    mustafar_package, compression = ref_modules
    torch.manual_seed(1)
    Bkv, T, D = 6, 512, 128
    for conv in (compression.convert_key_batched, compression.convert_value_batched):
        old = compression.prune_magnitude(torch.randn(Bkv, T, D, device=DEV).half(), 0.7)
        new = compression.prune_magnitude(torch.randn(Bkv, 256, D, device=DEV).half(), 0.7)
        _, _, nzs = conv(old)
        _, _, new_nzs = conv(new)
        packed = torch.cat(nzs)                                                            # model :274
        assert packed.data_ptr() == nzs[0].data_ptr() and packed.numel() == sum(n.numel() for n in nzs)
        before = torch.cuda.memory_allocated()
        for _ in range(3):
            assert torch.cat(nzs).data_ptr() == packed.data_ptr()
        assert torch.cuda.memory_allocated() <= before, "torch.cat of the pieces allocated something"
        want = [torch.cat([nzs[b].clone(), new_nzs[b].clone()], dim=0) for b in range(Bkv)]   # plain tensors: the reference's result
        merged = [torch.cat([nzs[b], new_nzs[b]], dim=0) for b in range(Bkv)]                # model :368
        for a, b in zip(merged, want):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        whole = torch.cat(merged)
        assert whole.data_ptr() == merged[0].data_ptr() and torch.equal(whole.view(torch.int16), torch.cat(want).view(torch.int16))
