"""GPU: the additive attention mask of the reference hook (models/llama_mustafar_kernel.py:293-301: `attn_weights +
attention_mask`, then `torch.max(., finfo.min)`, then the fp32 softmax) on the FUSED entry point.  The non-flash model
always passes a 4-D mask (:723-728), so this is the form the real hook reaches.  Held against (a) the unfused call
sequence with the same mask (same arithmetic, PyTorch glue) and (b) dense fp32 attention over the oracle-pruned K/V with
the masked columns removed.  Tolerance: fp16 (rtol 4e-3, atol 2e-3), as for the unmasked hook tests."""
import math

import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NEG = torch.finfo(torch.float16).min


def _dense_masked(q, K_all, V_all, C, ks, vs, groups, mask):
    K, V = K_all.clone(), V_all.clone()
    if C:
        K[:, :, :C] = torch.from_numpy(orc.prune_magnitude(K_all[:, :, :C].cpu().numpy(), ks)).to(K.device)
        V[:, :, :C] = torch.from_numpy(orc.prune_magnitude(V_all[:, :, :C].cpu().numpy(), vs)).to(V.device)
    Kr = K.float().repeat_interleave(groups, dim=1)
    Vr = V.float().repeat_interleave(groups, dim=1)
    s = torch.matmul(q.float(), Kr.transpose(2, 3)) / math.sqrt(q.shape[-1])
    s = s.masked_fill(mask < 0, float("-inf"))            # a column masked with finfo.min carries no weight
    return torch.matmul(torch.softmax(s, -1), Vr)


def _left_padding_mask(bsz, kv_len, pads):
    m = torch.zeros((bsz, 1, 1, kv_len), dtype=torch.float16, device=DEV)
    for b, p in enumerate(pads):
        m[b, :, :, :p] = NEG
    return m


@pytest.fixture
def structure(request):
    """Both structures of the fused entry point take the mask: the one-pass launch (1) and the two-launch form (0)."""
    from mustafar_amd import _lib
    lib = _lib.load()
    assert lib.mustafar_set_onepass(request.param) == 0
    yield request.param
    assert lib.mustafar_set_onepass(2) == 0


@pytest.mark.parametrize("structure", [0, 1], indirect=True)
@pytest.mark.parametrize("arena", [False, True])
@pytest.mark.parametrize("hq,hkv", [(8, 2), (4, 4), (8, 4)])
def test_fused_decode_with_left_padding_mask(arena, hq, hkv, structure):
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(3)
    bsz, D, L0, steps = 3, 128, 300, 9
    pads = (0, 37, 270)      # batch 2 masks the whole compressed part and some of the window
    cfg_f = MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, k_sparsity=0.7, v_sparsity=0.7, api="fused", arena=arena)
    cfg_n = MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, k_sparsity=0.7, v_sparsity=0.7, api="native")
    fused, native = MustafarAttention(cfg_f), MustafarAttention(cfg_n)
    groups = hq // hkv
    K_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    past_f = fused.build_cache(K_all.clone(), V_all.clone())
    past_n = native.build_cache(K_all.clone(), V_all.clone())
    for step in range(steps):     # kv_len runs over 301..309: every alignment of the mask rows
        qn = torch.randn(bsz, hq, 1, D, device=DEV).half()
        kn = torch.randn(bsz, hkv, 1, D, device=DEV).half()
        vn = torch.randn(bsz, hkv, 1, D, device=DEV).half()
        K_all, V_all = torch.cat([K_all, kn], 2), torch.cat([V_all, vn], 2)
        mask = _left_padding_mask(bsz, L0 + step + 1, pads)
        out_f, past_f = fused.decode(qn, kn, vn, past_f, attention_mask=mask)
        out_n, past_n = native.decode(qn, kn, vn, past_n, attention_mask=mask)
        torch.testing.assert_close(out_f.float(), out_n.float(), rtol=2e-3, atol=1e-3)     # same arithmetic, two call sequences
        want = _dense_masked(qn, K_all, V_all, 256, 0.7, 0.7, groups, mask)
        torch.testing.assert_close(out_f.float(), want, rtol=4e-3, atol=2e-3)
    # an all-zero mask changes nothing
    qn, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    zero = torch.zeros((bsz, 1, 1, past_f[5] + 1), dtype=torch.float16, device=DEV)
    p1 = (past_f[0], past_f[1].clone(), past_f[2], past_f[3].clone(), past_f[4], past_f[5])   # (no trigger fires here: the compressed part is shared)
    a, _ = fused.decode(qn, kn, vn, p1, attention_mask=zero)
    b, _ = fused.decode(qn, kn, vn, past_f)
    assert torch.equal(a, b)


def test_mask_shape_errors_match_the_reference():
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    attn = MustafarAttention(MustafarConfig(num_attention_heads=4, num_key_value_heads=2, api="fused"))
    K = torch.randn(1, 2, 300, 128, device=DEV).half()
    past = attn.build_cache(K.clone(), K.clone())
    q, kn, vn = (torch.randn(1, h, 1, 128, device=DEV).half() for h in (4, 2, 2))
    with pytest.raises(ValueError, match="Attention mask should be of size"):          # model :294-297
        attn.decode(q, kn, vn, past, attention_mask=torch.zeros((1, 1, 1, 300), dtype=torch.float16, device=DEV))
    with pytest.raises(RuntimeError, match="float16"):
        attn.decode(q, kn, vn, past, attention_mask=torch.zeros((1, 1, 1, 301), dtype=torch.float32, device=DEV))


@pytest.mark.parametrize("structure", [0, 1], indirect=True)
def test_masked_decode_with_rows_longer_than_32768(structure):
    """The streaming softmax form (T > 32768) applies the mask too."""
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(5)
    bsz, hq, hkv, D, L0 = 1, 4, 1, 128, 33024 + 40
    attn = MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, api="fused"))
    K = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    past = attn.to_fused(attn.build_cache(K.clone(), V.clone()))
    qn, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    K, V = torch.cat([K, kn], 2), torch.cat([V, vn], 2)
    mask = _left_padding_mask(bsz, L0 + 1, (20001,))
    out, past = attn.decode(qn, kn, vn, past, attention_mask=mask)
    want = _dense_masked(qn, K, V, 33024, 0.7, 0.7, hq // hkv, mask)
    torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=2e-3)


@pytest.mark.parametrize("structure", [0, 1], indirect=True)
def test_masked_decode_under_graph_replay(structure):
    """A captured step with a mask buffer as wide as the window capacity: each replay reads kv_len columns."""
    from mustafar_amd import _lib
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(9)
    bsz, hq, hkv, D, L0 = 2, 8, 2, 128, 300
    attn = MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, api="fused", arena=True))
    K_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    past = attn.to_fused(attn.build_cache(K_all.clone(), V_all.clone()))
    width = 256 + past[1].cap
    mask = torch.zeros((bsz, 1, 1, width), dtype=torch.float16, device=DEV)
    mask[1, :, :, :100] = NEG
    q, kn, vn = (torch.zeros(bsz, h, 1, D, device=DEV, dtype=torch.float16) for h in (hq, hkv, hkv))
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    lib = _lib.load()
    warm = (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5])
    attn.decode_fused(q, kn, vn, warm, attention_mask=mask[..., :L0 + 1].contiguous())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, _ = attn.decode_fused(q, kn, vn, past, step_counter=counter, attention_mask=mask)
        _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream().cuda_stream, counter.data_ptr(), 1), "counter")
    for step in range(5):
        qn, k1, v1 = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
        q.copy_(qn); kn.copy_(k1); vn.copy_(v1)
        K_all, V_all = torch.cat([K_all, k1], 2), torch.cat([V_all, v1], 2)
        g.replay()
        want = _dense_masked(qn, K_all, V_all, 256, 0.7, 0.7, hq // hkv, mask[..., :L0 + step + 1])
        torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=2e-3)


@pytest.mark.parametrize("structure", [0, 1], indirect=True)
def test_replayed_mask_must_cover_the_window_capacity(structure):
    """Under graph replay the kernels read mask columns up to compressed_length + window CAPACITY as the window grows: a mask of
    exactly that length is accepted and right on every replay up to a full window, a shorter one is refused on the host
    (before round 3 only `compressed_length + 1` columns were required and later replays read past the row)."""
    from mustafar_amd import _lib
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(21)
    bsz, hq, hkv, D, L0 = 2, 8, 2, 128, 300
    attn = MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, api="fused", arena=True))
    K_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    past = attn.to_fused(attn.build_cache(K_all.clone(), V_all.clone()))
    cap = past[1].cap
    q, kn, vn = (torch.zeros(bsz, h, 1, D, device=DEV, dtype=torch.float16) for h in (hq, hkv, hkv))
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    short = torch.zeros((bsz, 1, 1, 256 + cap - 1), dtype=torch.float16, device=DEV)
    with pytest.raises(ValueError, match="Attention mask should be of size"):
        attn.decode_fused(q, kn, vn, past, step_counter=counter, attention_mask=short)
    mask = torch.zeros((bsz, 1, 1, 256 + cap), dtype=torch.float16, device=DEV)     # exactly compressed_length + capacity
    mask[0, :, :, 10:60] = NEG
    mask[1, :, :, 256 + 50:256 + 70] = NEG                                              # columns the window only reaches later
    lib = _lib.load()
    warm = (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5])
    attn.decode_fused(q, kn, vn, warm, attention_mask=mask[..., :L0 + 1].contiguous())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, _ = attn.decode_fused(q, kn, vn, past, step_counter=counter, attention_mask=mask)
        _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream().cuda_stream, counter.data_ptr(), 1), "counter")
    steps = cap - (L0 - 256)            # until the window is full: the last replay reads the mask's last column
    for step in range(steps):
        qn, k1, v1 = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
        q.copy_(qn); kn.copy_(k1); vn.copy_(v1)
        K_all, V_all = torch.cat([K_all, k1], 2), torch.cat([V_all, v1], 2)
        g.replay()
        if step % 7 == 0 or step >= steps - 2:
            want = _dense_masked(qn, K_all, V_all, 256, 0.7, 0.7, hq // hkv, mask[..., :L0 + step + 1])
            torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=2e-3)
    assert K_all.shape[2] == 256 + cap
