"""GPU: tools/mem_spd.py, the mem_spd_test.py-shaped harness (prefill, then decode steps across the 256-token triggers,
1 warm-up + timed repeats, ms per generate + peak memory).  The reference script (mem_spd_test.py:81-96) only times; this
test holds what the harness times to the dense answer: the final decode step of every call sequence -- fused (eager and
graph-replayed), native, reference -- against fp32 attention over K / V pruned by the CPU ORACLE's prune rule, tracked
alongside the generate (prompt rows, then the appended decode rows; everything in front of the final compressed length
pruned, the window dense), through BOTH triggers of a 300 + 600-token run.  fp16 tolerance: rtol 4e-3, atol 2e-3."""
import importlib.util
import math
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("mem_spd", os.path.join(ROOT, "tools", "mem_spd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _dense_final(ms, gen, prompt, steps):
    """fp32 attention of the LAST layer's final decode step over the oracle-pruned K / V of the whole generate."""
    l = gen.a.layers - 1
    R = ms.GROUP_SIZE
    C = max(0, ((prompt - R) // 256) * 256)                   # prefill (model :416)
    kv = prompt
    for _ in range(steps):                                      # the trigger rule of the decode loop (model :324, :398)
        kv += 1
        if (kv - R - C) % 256 == 0 and kv - C >= 256:
            C += 256
    K = torch.cat([gen.pk, gen.dk[l].expand(-1, -1, steps, -1)], 2).cpu().numpy()
    V = torch.cat([gen.pv, gen.dv[l].expand(-1, -1, steps, -1)], 2).cpu().numpy()
    B, Hkv, L, D = K.shape
    assert L == kv
    K[:, :, :C] = orc.prune_magnitude(np.ascontiguousarray(K[:, :, :C]).reshape(B * Hkv, C, D), ms.K_SPARSITY).reshape(B, Hkv, C, D)
    V[:, :, :C] = orc.prune_magnitude(np.ascontiguousarray(V[:, :, :C]).reshape(B * Hkv, C, D), ms.V_SPARSITY).reshape(B, Hkv, C, D)
    groups = ms.Q_HEADS // ms.KV_HEADS
    q = gen.dq[l].cpu().numpy().astype(np.float64).reshape(B, Hkv, groups, D)
    s = np.einsum("bhgd,bhtd->bhgt", q, K.astype(np.float64)) / math.sqrt(D)
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    out = np.einsum("bhgt,bhtd->bhgd", p, V.astype(np.float64)).reshape(B, ms.Q_HEADS, 1, D)
    return torch.from_numpy(out).float(), C, kv


def test_mem_spd_generate_matches_dense_through_both_triggers():
    ms = _load()
    prompt, steps = 300, 600                                     # the reference's run (mem_spd_test.py:72-74) at a small batch
    common = ["--batch", "2", "--layers", "2", "--prompt-length", str(prompt), "--output-length", str(steps), "--repeats", "1"]
    keep = {}
    res, outs = ms.main(["--api", "fused", "native", "reference"] + common, keep=keep)
    assert [r["api"] for r in res] == ["fused", "native", "reference"]
    want, C, kv = _dense_final(ms, keep["native"], prompt, steps)
    assert (C, kv) == (768, 900)
    for r in res:
        assert r["triggers_per_generate"] == 2 and r["final_compressed_tokens"] == C and r["final_kv_seq_len"] == kv
        assert r["ms_per_generate_avg"] > 0 and r["peak_mem_gb"] > 0 and len(r["ms_per_generate"]) == 1
    for api in ("fused", "native", "reference"):
        torch.testing.assert_close(outs[api].float().cpu(), want, rtol=4e-3, atol=2e-3, msg=lambda m, api=api: f"{api}: {m}")
    keep_g = {}
    res_g, outs_g = ms.main(["--api", "fused", "--graph"] + common, keep=keep_g)
    assert res_g[0]["api"] == "fused+graph" and res_g[0]["triggers_per_generate"] == 2 and res_g[0]["final_kv_seq_len"] == kv
    assert res_g[0]["final_compressed_tokens"] == C
    torch.testing.assert_close(outs_g["fused"].float().cpu(), want, rtol=4e-3, atol=2e-3)
    # the generators of the runs were seeded alike: the dense answer above is the answer of every one of them
    for g in list(keep.values()) + list(keep_g.values()):
        assert torch.equal(g.pk, keep["native"].pk) and torch.equal(g.dq[-1], keep["native"].dq[-1])


def test_mem_spd_harness_smoke_across_one_trigger():
    ms = _load()
    # prompt 500 -> 256 compressed + 244 in the window: the trigger fires at decode step 44 of 50
    prompt, steps = 500, 50
    common = ["--batch", "2", "--layers", "2", "--prompt-length", str(prompt), "--output-length", str(steps), "--repeats", "1"]
    keep = {}
    res, outs = ms.main(["--api", "fused", "reference"] + common, keep=keep)
    want, C, kv = _dense_final(ms, keep["fused"], prompt, steps)
    assert (C, kv) == (512, 550)
    for r in res:
        assert r["triggers_per_generate"] == 1 and r["final_compressed_tokens"] == 512 and r["final_kv_seq_len"] == 550
    for api in ("fused", "reference"):
        torch.testing.assert_close(outs[api].float().cpu(), want, rtol=4e-3, atol=2e-3)


def test_checkpoint_must_be_a_local_directory():
    ms = _load()
    with pytest.raises(SystemExit, match="local directory"):
        ms.main(["--checkpoint", "meta-llama/Meta-Llama-3-8B-Instruct"])
