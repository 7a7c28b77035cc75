"""GPU: the hook mirror (prefill + decode incl. a 256-token compression trigger) against dense attention over the
pruned-but-dense K/V -- the relationship between the reference's kernel model (llama_mustafar_kernel.py) and its
dense accuracy model (llama_mustafar_Kt_Mag_Vt_Mag.py:873, :963, :974).  Tolerance: the scale-relative
comparator of the bench-shape suite (tests/util.py: 3 fp16 ulp of max|out| + 1e-4; round 4 used rtol 4e-3, atol 2e-3 here)."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.util import DENSE_ULPS, excess

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dense_reference(q, K_all, V_all, C, ks, vs, groups):
    """softmax(q K^T / sqrt(d)) V in fp32 with tokens [:C] pruned by the reference rule (CPU oracle)."""
    K = K_all.clone()
    V = V_all.clone()
    if C:
        K[:, :, :C] = torch.from_numpy(orc.prune_magnitude(K_all[:, :, :C].cpu().numpy(), ks)).to(K.device)
        V[:, :, :C] = torch.from_numpy(orc.prune_magnitude(V_all[:, :, :C].cpu().numpy(), vs)).to(V.device)
    Kr = K.float().repeat_interleave(groups, dim=1)
    Vr = V.float().repeat_interleave(groups, dim=1)
    s = torch.matmul(q.float(), Kr.transpose(2, 3)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(s, -1), Vr)


@pytest.mark.parametrize("api", ["reference", "native", "fused"])
@pytest.mark.parametrize("hq,hkv", [(8, 2), (4, 4), (8, 4)])
def test_prefill_then_decode_matches_dense(api, hq, hkv):
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(42)
    bsz, D, L0, steps = 2, 128, 300, 262
    cfg = MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, k_sparsity=0.7, v_sparsity=0.5, api=api)
    attn = MustafarAttention(cfg)
    groups = hq // hkv
    q = torch.randn(bsz, hq, L0, D, device=DEV).half()
    K_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    out, past = attn.prefill(q, K_all.clone(), V_all.clone())
    assert past[4] == 256 and past[5] == L0 and past[1].shape[2] == L0 - 256
    assert out.shape == (bsz, hq, L0, D)
    fired = 0
    for step in range(steps):
        qn = torch.randn(bsz, hq, 1, D, device=DEV).half()
        kn = torch.randn(bsz, hkv, 1, D, device=DEV).half()
        vn = torch.randn(bsz, hkv, 1, D, device=DEV).half()
        C_before = past[4]
        K_all = torch.cat([K_all, kn], 2)
        V_all = torch.cat([V_all, vn], 2)
        out, past = attn.decode(qn, kn, vn, past)
        if step % 20 == 0 or past[4] != C_before or step == steps - 1:
            want = _dense_reference(qn, K_all, V_all, C_before, cfg.k_sparsity, cfg.v_sparsity, groups)
            assert excess(out, want, DENSE_ULPS) <= 1.0
        fired += past[4] != C_before
        assert past[5] == L0 + step + 1
        wlen = past[1].len if api == "fused" else past[1].shape[2]
        assert wlen == past[5] - past[4]
    assert fired == 1 and past[4] == 512          # one 256-token trigger crossed (model :324)
    # cache tuple keeps the reference layout (model :445, :332): [bitmaps, idxs, list-of-streams, nz_offset]
    kc = past[0]
    assert kc[0].dtype == torch.int64 and kc[1].dtype == torch.int32 and kc[3].dtype == torch.int32
    assert isinstance(kc[2], list) and len(kc[2]) == bsz * hkv
    assert kc[0].numel() == bsz * hkv * 2 * 512 and kc[1].numel() == bsz * hkv * (2 * 512 + 1)


@pytest.mark.parametrize("api", ["native", "fused"])
def test_decode_without_compressed_part(api):
    """compressed_length == 0 branch (model :280-282, :318-320): short prompt stays dense until the first trigger."""
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(1)
    attn = MustafarAttention(MustafarConfig(num_attention_heads=4, num_key_value_heads=2, api=api))
    q = torch.randn(1, 4, 40, 128, device=DEV).half()
    k = torch.randn(1, 2, 40, 128, device=DEV).half()
    v = torch.randn(1, 2, 40, 128, device=DEV).half()
    _, past = attn.prefill(q, k, v)
    assert past[0] is None and past[4] == 0
    K_all, V_all = k, v
    for _ in range(3):
        qn, kn, vn = (torch.randn(1, h, 1, 128, device=DEV).half() for h in (4, 2, 2))
        K_all, V_all = torch.cat([K_all, kn], 2), torch.cat([V_all, vn], 2)
        out, past = attn.decode(qn, kn, vn, past)
        want = _dense_reference(qn, K_all, V_all, 0, 0.7, 0.7, 2)
        assert excess(out, want, DENSE_ULPS) <= 1.0


def test_fused_decode_under_graph_replay():
    """A captured hipGraph of one fused decode step, replayed while a device counter grows the window."""
    from mustafar_amd import _lib
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(7)
    bsz, hq, hkv, D, L0 = 1, 8, 2, 128, 300
    cfg = MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, k_sparsity=0.7, v_sparsity=0.7, api="fused")
    attn = MustafarAttention(cfg)
    K_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    past = attn.to_fused(attn.build_cache(K_all.clone(), V_all.clone()))
    q, kn, vn = (torch.zeros(bsz, h, 1, D, device=DEV, dtype=torch.float16) for h in (hq, hkv, hkv))
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    lib = _lib.load()
    # warm-up on a private copy (allocates the scratch buffers outside the capture)
    warm = (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5])
    attn.decode_fused(q, kn, vn, warm)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, _ = attn.decode_fused(q, kn, vn, past, step_counter=counter)
        _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream().cuda_stream, counter.data_ptr(), 1), "counter")
    for step in range(12):
        qn, k1, v1 = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
        q.copy_(qn); kn.copy_(k1); vn.copy_(v1)
        K_all, V_all = torch.cat([K_all, k1], 2), torch.cat([V_all, v1], 2)
        g.replay()
        want = _dense_reference(qn, K_all, V_all, 256, 0.7, 0.7, hq // hkv)
        assert excess(out, want, DENSE_ULPS) <= 1.0
    assert int(counter.item()) == 12
    past = attn.advance(past, 12)
    assert past[1].len == L0 - 256 + 12 and past[5] == L0 + 12
    torch.testing.assert_close(past[1].view()[:, :, -1], K_all[:, :, -1])
    # and the eager path continues from the advanced state
    qn, k1, v1 = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    K_all, V_all = torch.cat([K_all, k1], 2), torch.cat([V_all, v1], 2)
    out2, past = attn.decode_fused(qn, k1, v1, past)
    assert excess(out2, _dense_reference(qn, K_all, V_all, 256, 0.7, 0.7, hq // hkv), DENSE_ULPS) <= 1.0


_ALT_FORMS = r"""
import math, sys, torch
sys.path.insert(0, {root!r})
from mustafar_amd.hook import MustafarAttention, MustafarConfig
from tests.test_gpu_hook import _dense_reference
from tests.util import DENSE_ULPS, excess
torch.manual_seed(11)
dev, bsz, hq, hkv, D, L0 = "cuda:0", 2, 8, 2, 128, 600
cfg = MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, api="fused")
attn = MustafarAttention(cfg)
K = torch.randn(bsz, hkv, L0, D, device=dev).half(); V = torch.randn(bsz, hkv, L0, D, device=dev).half()
past = attn.to_fused(attn.build_cache(K.clone(), V.clone()))
for step in range(6):
    qn, kn, vn = (torch.randn(bsz, h, 1, D, device=dev).half() for h in (hq, hkv, hkv))
    K, V = torch.cat([K, kn], 2), torch.cat([V, vn], 2)
    C = past[4]
    out, past = attn.decode(qn, kn, vn, past)
    want = _dense_reference(qn, K, V, C, 0.7, 0.7, hq // hkv)
    assert excess(out, want, DENSE_ULPS) <= 1.0
print("alt-form ok")
"""


@pytest.mark.parametrize("env", [{"MUSTAFAR_ONEPASS": "0"}, {"MUSTAFAR_ONEPASS": "1"}, {"MUSTAFAR_ONEPASS": "1", "MUSTAFAR_FMA_ENGINE": "mfma"},
                                 {"MUSTAFAR_ONEPASS": "1", "MUSTAFAR_ONEPASS_WGS": "7"}, {"MUSTAFAR_ONEPASS": "1", "MUSTAFAR_SB": "0"},
                                 {"MUSTAFAR_ONEPASS": "1", "MUSTAFAR_SB": "0", "MUSTAFAR_FMA_ENGINE": "mfma"},
                                 {"MUSTAFAR_ONEPASS": "0", "MUSTAFAR_WINDOW": "rows"}, {"MUSTAFAR_ONEPASS": "0", "MUSTAFAR_WINDOW": "key"},
                                 {"MUSTAFAR_ONEPASS": "0", "MUSTAFAR_VALUE_SPLIT": "1"}, {"MUSTAFAR_ONEPASS": "0", "MUSTAFAR_KEY_LEAN": "0", "MUSTAFAR_KEY_SPLIT": "1"},
                                 {"MUSTAFAR_ONEPASS": "0", "MUSTAFAR_KEY_LEAN": "0", "MUSTAFAR_KEY_SPLIT": "2", "MUSTAFAR_FMA_ENGINE": "mfma"},
                                 {"MUSTAFAR_ONEPASS": "0", "MUSTAFAR_KEY_LEAN": "0"}, {"MUSTAFAR_ONEPASS": "0", "MUSTAFAR_VALUE_LEAN": "1"},
                                 {"MUSTAFAR_ONEPASS": "0", "MUSTAFAR_WINDOW_POS": "first"}, {"MUSTAFAR_ONEPASS": "0", "MUSTAFAR_WINDOW_POS": "last"}])
def test_alternative_kernel_forms_in_a_child_process(env):
    """The launch-shape switches are read once per process: each form -- the one-pass launch and the two-launch form with
    its window / split / engine variants -- decodes a few steps in a child process and is held against dense attention."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", _ALT_FORMS.format(root=root)], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "alt-form ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("hq,hkv", [(8, 8), (8, 4), (8, 2)])
@pytest.mark.parametrize("L0", [1312, 2100])
def test_four_blocks_per_workgroup_for_every_group_count(hq, hkv, L0):
    """The one-pass launch gives a workgroup four 64-token blocks -- each pair of waves two consecutive blocks as ONE pipeline, block B's e
    segments G x 64 halfs behind block A's -- only when that leaves >= 768 workgroups (1024 when the bug was shipped): MHA and GQA-2 reach that shape at 8k x batch 8, far
    above the sizes of this suite (round 5 shipped a build whose G < 4 launches read block B's e at the G = 4 offset; bench-shape tests
    are GQA-4 or small).  mustafar_tune(1, 2) forces the shape at a size dense attention checks in a second."""
    from mustafar_amd import _lib
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    L = _lib.load()
    torch.manual_seed(5)
    bsz, D = 2, 128
    attn = MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, api="fused"))
    K = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    past = attn.to_fused(attn.build_cache(K.clone(), V.clone()))
    assert L.mustafar_tune(1, 2) == 0
    try:
        for _ in range(3):
            qn, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
            K, V = torch.cat([K, kn], 2), torch.cat([V, vn], 2)
            C = past[4]
            out, past = attn.decode(qn, kn, vn, past)
            assert C >= 1024 and excess(out, _dense_reference(qn, K, V, C, 0.7, 0.7, hq // hkv), DENSE_ULPS) <= 1.0
    finally:
        L.mustafar_tune(1, 0)


@pytest.mark.parametrize("hq,hkv", [(8, 2), (8, 8)])
def test_both_row_kernels_of_short_rows(hq, hkv):
    """Rows of <= 64 slabs are merged by onepass_finish1_kernel (one thread per channel, weights by v_readlane; round 5); mustafar_tune(10, 0)
    keeps them on the 256-thread kernel (its one-slab-per-lane text, which no default launch reaches any more).  Both against dense attention,
    and against each other within the rounding of one fp16 output."""
    from mustafar_amd import _lib
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    L = _lib.load()
    torch.manual_seed(9)
    bsz, D, L0 = 2, 128, 1600
    attn = MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, api="fused"))
    K = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    past0 = attn.to_fused(attn.build_cache(K.clone(), V.clone()))
    qn, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    K, V = torch.cat([K, kn], 2), torch.cat([V, vn], 2)
    want = _dense_reference(qn, K, V, past0[4], 0.7, 0.7, hq // hkv)
    outs = []
    try:
        for form in (1, 0):
            assert L.mustafar_tune(10, form) == 0
            past = (past0[0], past0[1].clone(), past0[2], past0[3].clone(), past0[4], past0[5])
            out, _ = attn.decode(qn, kn, vn, past)
            assert excess(out, want, DENSE_ULPS) <= 1.0
            outs.append(out.float())
    finally:
        L.mustafar_tune(10, 1)
    assert (outs[0] - outs[1]).abs().max().item() <= 2 ** -10 * want.abs().max().item() + 1e-6


def test_fused_decode_with_rows_longer_than_32768():
    """Compressed length 33024 > 32768: the softmax runs in its streaming form (three passes over the row)."""
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(5)
    bsz, hq, hkv, D, L0 = 1, 4, 1, 128, 33024 + 40
    attn = MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, api="fused"))
    K = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    past = attn.to_fused(attn.build_cache(K.clone(), V.clone()))
    assert past[4] == 33024
    for _ in range(2):
        qn, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
        K, V = torch.cat([K, kn], 2), torch.cat([V, vn], 2)
        out, past = attn.decode(qn, kn, vn, past)
        want = _dense_reference(qn, K, V, 33024, 0.7, 0.7, hq // hkv)
        assert excess(out, want, DENSE_ULPS) <= 1.0


def test_fused_decode_on_two_streams_at_once():
    """Two streams, each with its own cache and its own scratch (hook.py keeps score scratch and slab workspace per device and
    stream): the calls are enqueued alternately without a sync in between and must give what the same calls give one after the
    other."""
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(5)
    cfg = MustafarConfig(num_attention_heads=8, num_key_value_heads=2, k_sparsity=0.7, v_sparsity=0.7, residual_length=32, api="fused", arena=True)
    attn = MustafarAttention(cfg)
    dev = "cuda:0"
    K = [torch.randn(2, 2, 1056, 128, device=dev).half() for _ in range(2)]
    V = [torch.randn(2, 2, 1056, 128, device=dev).half() for _ in range(2)]
    steps = [[tuple(torch.randn(2, h, 1, 128, device=dev).half() for h in (8, 2, 2)) for _ in range(6)] for _ in range(2)]

    def run(parallel):
        pasts = [attn.to_fused(attn.build_cache(K[i], V[i])) for i in range(2)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()] if parallel else [torch.cuda.current_stream()] * 2
        torch.cuda.synchronize()
        outs = [[], []]
        for n in range(6):
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    q, k, v = steps[i][n]
                    o, pasts[i] = attn.decode(q, k, v, pasts[i])
                    outs[i].append(o)
        torch.cuda.synchronize()
        return [torch.stack(o) for o in outs]

    serial, parallel = run(False), run(True)
    for a, b in zip(serial, parallel):
        assert torch.equal(a, b)
