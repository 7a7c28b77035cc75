"""GPU: the engine / structure choice of the fused entry point travels in the call (`flags`, MustafarConfig.engine / .structure), not in
process state; what the dot2 engine does to fp16 subnormals is what the header says and nothing more; and the e round trip
of the one-pass launches (vector stores, scalar loads of the same scratch, launch after launch) holds under graph replay.

  * two MustafarAttention objects with different engines / structures in one process each run their own kernels -- seen from the
    outside: the profile records tell a one-pass launch from two launches, and an input whose scores rest on fp16-subnormal K
    entries alone tells the exact v_fma_mix engine from v_dot2 (which counts them as zero: include/mustafar_hip.h);
  * on ordinary data the three engines agree within fp16 (each is held to dense fp32 attention over oracle-pruned K / V);
  * a captured step replayed >= 32 times with two alternating queries on ONE score scratch: every replay equals dense
    attention for the query it was given (a stale scalar-cache line of the previous replay's e values would show at once).
"""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dense(q, K_all, V_all, C, s, groups):
    K, V = K_all.clone(), V_all.clone()
    if C:
        K[:, :, :C] = torch.from_numpy(orc.prune_magnitude(K_all[:, :, :C].cpu().numpy(), s)).to(K.device)
        V[:, :, :C] = torch.from_numpy(orc.prune_magnitude(V_all[:, :, :C].cpu().numpy(), s)).to(V.device)
    Kr = K.double().repeat_interleave(groups, dim=1)
    Vr = V.double().repeat_interleave(groups, dim=1)
    sc = torch.matmul(q.double(), Kr.transpose(2, 3)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(sc, -1), Vr).float()


def _attn(hq, hkv, **kw):
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    return MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, k_sparsity=0.7, v_sparsity=0.7,
                                            api="fused", arena=True, **kw))


def _profiled(lib, fn):
    from mustafar_amd import _lib
    _lib.check(lib.mustafar_profile_begin(4), "profile_begin")
    out = fn()
    ku, vu, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    _lib.check(lib.mustafar_profile_end(ctypes.byref(ku), ctypes.byref(vu), ctypes.byref(n)), "profile_end")
    return out, ku.value, vu.value, n.value


def test_two_instances_each_run_their_own_structure_and_engine():
    from mustafar_amd import _lib
    lib = _lib.load()
    torch.manual_seed(11)
    bsz, hq, hkv, D, L0 = 2, 8, 2, 128, 1056      # 1024 compressed tokens + 32 in the window
    K = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    q, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    want = _dense(q, torch.cat([K, kn], 2), torch.cat([V, vn], 2), 1024, 0.7, hq // hkv)
    # (engine that must run, structure, one-pass form) by instance; mustafar_last_decode_choice() reports what the call launched
    # (one-pass form 4 = round 6's small-launch kernel: this shape is 16 workgroups; the matrix-pipe engine keeps the super-block kernel, form 3)
    cases = (("dot2", "one_pass", 2 | 1 << 4 | 4 << 8), ("valu", "one_pass", 0 | 1 << 4 | 4 << 8), ("mfma", "one_pass", 1 | 1 << 4 | 3 << 8),
             ("valu", "two_launch", 0), ("mfma", "two_launch", 1), (None, None, 2 | 1 << 4 | 4 << 8))
    attns = [(_attn(hq, hkv, engine=e, structure=st), code) for e, st, code in cases]
    pasts = [a.to_fused(a.build_cache(K.clone(), V.clone())) for a, _ in attns]
    for rnd in range(2):                              # interleaved: no instance inherits what the previous call chose
        for (a, code), past in zip(attns, pasts):
            p = (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5])
            (out, _), ku, vu, n = _profiled(lib, lambda: a.decode(q, kn, vn, p))
            assert lib.mustafar_last_decode_choice() == code, (a.cfg.engine, a.cfg.structure, hex(lib.mustafar_last_decode_choice()))
            assert n == 1 and ku > 0 and (vu == 0) == bool(code & (1 << 4)), "the profile records show the other structure"
            torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=2e-3)
    # the process defaults are untouched by per-call choices
    assert lib.mustafar_get_fma_engine() == 2 and lib.mustafar_get_onepass() == 2


def test_dot2_error_on_subnormal_inputs_is_bounded_by_a_flush():
    """Every K element an fp16 SUBNORMAL, large queries: the scores of the compressed part rest on subnormal inputs alone.  The exact
    engine matches dense attention; the dot2 engine may lose such products (include/mustafar_hip.h) but never does worse than
    dropping all of them: it lies between the exact answer and the answer with the compressed part's K counted as zero."""
    torch.manual_seed(12)
    bsz, hq, hkv, D, L0 = 2, 8, 2, 128, 1056
    lim = 2.0 ** -14 - 2.0 ** -24
    Ks = (torch.randn(bsz, hkv, L0, D, device=DEV) * 2.0 ** -16).clamp(-lim, lim).half()
    V = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    q = (torch.randn(bsz, hq, 1, D, device=DEV) * 16384).clamp(-60000, 60000).half()
    kn = (torch.randn(bsz, hkv, 1, D, device=DEV) * 2.0 ** -16).clamp(-lim, lim).half()
    vn = torch.randn(bsz, hkv, 1, D, device=DEV).half()
    outs = {}
    for eng in ("dot2", "valu"):
        a = _attn(hq, hkv, engine=eng, structure="one_pass")
        out, _ = a.decode(q, kn, vn, a.to_fused(a.build_cache(Ks.clone(), V.clone())))
        outs[eng] = out.float()
    Kall, Vall = torch.cat([Ks, kn], 2), torch.cat([V, vn], 2)
    exact = _dense(q, Kall, Vall, 1024, 0.7, hq // hkv)
    torch.testing.assert_close(outs["valu"], exact, rtol=4e-3, atol=2e-3)
    Kz = Kall.clone()
    Kz[:, :, :1024] = 0
    flushed = _dense_zeroK(q, Kz, Vall, 1024, hq // hkv)
    worst_flush = float((flushed - exact).abs().max())
    assert worst_flush > 1e-2, "the input does not exercise the subnormal path"
    assert float((outs["dot2"] - exact).abs().max()) <= worst_flush + 2e-3


def _dense_zeroK(q, Kz, V_all, C, groups):
    """dense attention with V pruned over the compressed part and K given as is (its compressed rows are zero)."""
    V = V_all.clone()
    V[:, :, :C] = torch.from_numpy(orc.prune_magnitude(V_all[:, :, :C].cpu().numpy(), 0.7)).to(V.device)
    Kr = Kz.double().repeat_interleave(groups, dim=1)
    Vr = V.double().repeat_interleave(groups, dim=1)
    sc = torch.matmul(q.double(), Kr.transpose(2, 3)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(sc, -1), Vr).float()


@pytest.mark.parametrize("hq,hkv,L0", [(8, 2, 1056), (32, 8, 2080)])
def test_engines_agree_on_ordinary_data_and_small_weights_survive_dot2(hq, hkv, L0):
    """N(0,1) data has no fp16 subnormals among the kept elements: dot2, fma_mix and mfma all sit inside the fp16 band around
    dense attention.  One batch entry gets a sharply peaked row (one score 12 above the rest): its other weights are ~6e-6, fp16
    subnormals as e values -- the 2^15 scale of the dot2 form keeps them, so the output still matches."""
    torch.manual_seed(13)
    bsz, D = 2, 128
    C = ((L0 - 32) // 256) * 256
    K = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    q = torch.randn(bsz, hq, 1, D, device=DEV).half()
    # batch 1: every query head looks hard at token 100 of its kv-head (score ~ |k|^2 / sqrt(d) * 1.5 >> the others)
    Kp = torch.from_numpy(orc.prune_magnitude(K[:, :, :C].cpu().numpy(), 0.7)).to(DEV)
    q[1] = (Kp[1, :, 100].float() * 1.5).half().repeat_interleave(hq // hkv, dim=0)[:, None, :]
    kn, vn = (torch.randn(bsz, hkv, 1, D, device=DEV).half() for _ in range(2))
    want = _dense(q, torch.cat([K, kn], 2), torch.cat([V, vn], 2), C, 0.7, hq // hkv)
    got = {}
    for eng in ("dot2", "valu", "mfma"):
        a = _attn(hq, hkv, engine=eng, structure="one_pass")
        past = a.to_fused(a.build_cache(K.clone(), V.clone()))
        out, _ = a.decode(q, kn, vn, past)
        got[eng] = out.float()
        torch.testing.assert_close(got[eng], want, rtol=4e-3, atol=2e-3, msg=lambda m, eng=eng: f"{eng}: {m}")
    scale = float(want.abs().max())
    assert float((got["dot2"] - got["valu"]).abs().max()) <= 2 * 2.0 ** -11 * scale + 1e-4


@pytest.mark.parametrize("hq,hkv,L0,bsz", [(8, 2, 1056, 2), (32, 32, 3872, 1)])
def test_replays_with_alternating_queries_on_one_scratch(hq, hkv, L0, bsz):
    """The e round trip (spmv.hip: INVARIANT at decode_onepass_lean_kernel / the pair form of round 2 for G < 4): a captured
    one-pass step replayed 34 times, the query alternating between two very different ones; the score scratch, the slab
    workspace and every address are the same in all replays.  (8, 2): GQA-4, the lean pair kernel; (32, 32): c2's geometry, G = 1.)"""
    from mustafar_amd import _lib
    lib = _lib.load()
    torch.manual_seed(17)
    D, C = 128, ((L0 - 32) // 256) * 256
    a = _attn(hq, hkv, structure="one_pass")
    K_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    past = a.to_fused(a.build_cache(K_all.clone(), V_all.clone()))
    # two unrelated queries (the second one sharper: x1.5; beyond that the fp16 rounding of the scores themselves, SpMM_Kernel.cuh:418,
    # outgrows the fp16 band of the outputs): a replay that picked up the previous replay's e values would be off by O(1)
    qs = [torch.randn(bsz, hq, 1, D, device=DEV).half(), (torch.randn(bsz, hq, 1, D, device=DEV) * 1.5).half()]
    Kp, Vp = K_all.clone(), V_all.clone()        # the dense answer's K / V: compressed part pruned once (oracle rule), the rest as is
    Kp[:, :, :C] = torch.from_numpy(orc.prune_magnitude(K_all[:, :, :C].cpu().numpy(), 0.7)).to(DEV)
    Vp[:, :, :C] = torch.from_numpy(orc.prune_magnitude(V_all[:, :, :C].cpu().numpy(), 0.7)).to(DEV)
    q, kn, vn = (torch.zeros(bsz, h, 1, D, device=DEV, dtype=torch.float16) for h in (hq, hkv, hkv))
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    warm = (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5])
    a.decode_fused(q, kn, vn, warm)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, _ = a.decode_fused(q, kn, vn, past, step_counter=counter)
        _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream().cuda_stream, counter.data_ptr(), 1), "counter")
    for step in range(34):
        k1, v1 = (torch.randn(bsz, hkv, 1, D, device=DEV).half() for _ in range(2))
        q.copy_(qs[step & 1]); kn.copy_(k1); vn.copy_(v1)
        Kp, Vp = torch.cat([Kp, k1], 2), torch.cat([Vp, v1], 2)
        g.replay()
        torch.testing.assert_close(out.float(), _dense(qs[step & 1], Kp, Vp, 0, 0.7, hq // hkv), rtol=4e-3, atol=2e-3,
                                   msg=lambda m, step=step: f"replay {step}: {m}")


def _left_pad_mask(bsz, kv_len, pads):
    m = torch.zeros((bsz, 1, 1, kv_len), dtype=torch.float16, device=DEV)
    for b, p in enumerate(pads):
        m[b, :, :, :p] = torch.finfo(torch.float16).min
    return m


@pytest.mark.parametrize("hq,hkv,eng", [(8, 2, "dot2"), (8, 2, "valu"), (8, 4, None), (8, 8, None), (32, 8, "dot2")])
def test_small_launch_kernel_against_the_super_block_kernel_and_dense(hq, hkv, eng):
    """Round 6, decode_onepass_small_kernel (spmv.hip): launches of two blocks per workgroup that do not put a wave on every SIMD.  Forced on
    (mustafar_tune(11, 2)) and off (11, 0: the super-block kernel) on the same cache: both inside the fp16 band around dense attention over
    oracle-pruned K / V, and within 2 ulp of the output scale of each other -- eager with and without a left-padding mask, through a 256-token
    trigger (the cache then holds an extent: the EXT instantiation) and as a captured graph replayed with two alternating queries (the e
    values cross the pair through LDS here; the window length grows under the replays)."""
    from mustafar_amd import _lib
    lib = _lib.load()
    torch.manual_seed(23)
    bsz, D, L0 = 2, 128, 1024 + 32 + 250                 # 1024 compressed tokens, a window six tokens short of the trigger
    C = 1024
    groups = hq // hkv
    K_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    Kp, Vp = K_all.clone(), V_all.clone()
    Kp[:, :, :C] = torch.from_numpy(orc.prune_magnitude(K_all[:, :, :C].cpu().numpy(), 0.7)).to(DEV)
    Vp[:, :, :C] = torch.from_numpy(orc.prune_magnitude(V_all[:, :, :C].cpu().numpy(), 0.7)).to(DEV)
    kw = {"structure": "one_pass"}
    if eng:
        kw["engine"] = eng
    outs = {}
    try:
        for form in (2, 0):
            assert lib.mustafar_tune(11, form) == 0
            a = _attn(hq, hkv, **kw)
            past = a.to_fused(a.build_cache(K_all.clone(), V_all.clone()))
            K, V = Kp.clone(), Vp.clone()
            torch.manual_seed(29)
            got = []
            for step in range(12):                       # the trigger fires at step 6: steps 7.. read the appended extent
                q, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
                K, V = torch.cat([K, kn], 2), torch.cat([V, vn], 2)
                mask = _left_pad_mask(bsz, K.shape[2], (0, 70)) if step % 3 == 2 else None
                out, past = a.decode(q, kn, vn, past, attention_mask=mask)
                assert (lib.mustafar_last_decode_choice() >> 8) & 15 == (4 if form else 3), "the launch form asked for did not run"
                if mask is not None:                     # batch 1: its first 70 columns carry no weight
                    want = torch.stack([_dense(q[b:b + 1], K[b:b + 1, :, p:], V[b:b + 1, :, p:], 0, 0.7, groups)[0] for b, p in enumerate((0, 70))])
                else:
                    want = _dense(q, K, V, 0, 0.7, groups)
                torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=2e-3, msg=lambda m, step=step, form=form: f"form {form} step {step}: {m}")
                got.append(out.float())
                if past[4] > C:                          # behind this step the hook pruned + compressed 256 more tokens (model :324): so does the dense reference
                    K[:, :, C:C + 256] = torch.from_numpy(orc.prune_magnitude(K[:, :, C:C + 256].cpu().numpy(), 0.7)).to(DEV)
                    V[:, :, C:C + 256] = torch.from_numpy(orc.prune_magnitude(V[:, :, C:C + 256].cpu().numpy(), 0.7)).to(DEV)
                    C += 256
            outs[form] = got
            C = 1024
        for x, y in zip(outs[2], outs[0]):
            scale = float(y.abs().max())
            assert float((x - y).abs().max()) <= 2 * 2.0 ** -11 * scale + 1e-4
        # ---- captured + replayed with alternating queries
        assert lib.mustafar_tune(11, 2) == 0
        a = _attn(hq, hkv, **kw)
        past = a.to_fused(a.build_cache(K_all[:, :, :1024 + 32].clone(), V_all[:, :, :1024 + 32].clone()))
        K, V = Kp[:, :, :1024 + 32].clone(), Vp[:, :, :1024 + 32].clone()
        qs = [torch.randn(bsz, hq, 1, D, device=DEV).half(), (torch.randn(bsz, hq, 1, D, device=DEV) * 1.5).half()]
        q, kn, vn = (torch.zeros(bsz, h, 1, D, device=DEV, dtype=torch.float16) for h in (hq, hkv, hkv))
        counter = torch.zeros(1, dtype=torch.int32, device=DEV)
        a.decode_fused(q, kn, vn, (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5]))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out, _ = a.decode_fused(q, kn, vn, past, step_counter=counter)
            _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream().cuda_stream, counter.data_ptr(), 1), "counter")
        for step in range(10):
            k1, v1 = (torch.randn(bsz, hkv, 1, D, device=DEV).half() for _ in range(2))
            q.copy_(qs[step & 1]); kn.copy_(k1); vn.copy_(v1)
            K, V = torch.cat([K, k1], 2), torch.cat([V, v1], 2)
            g.replay()
            torch.testing.assert_close(out.float(), _dense(qs[step & 1], K, V, 0, 0.7, groups), rtol=4e-3, atol=2e-3,
                                       msg=lambda m, step=step: f"replay {step}: {m}")
    finally:
        lib.mustafar_tune(11, 1)


@pytest.mark.parametrize("hq,hkv", [(8, 2), (8, 8)])
@pytest.mark.parametrize("kind", ["dense", "one_half", "zero_block"])
def test_one_pass_forms_on_extreme_tiles(hq, hkv, kind):
    """The fused launch on the tiles SURVEY 8d calls adversarial, for the three one-pass forms that serve ordinary shapes -- the small-launch kernel (round 6),
    the super-block kernel with two and with four blocks per workgroup: `dense` = nothing pruned (sparsity 0: every tile holds 64 non-zeros, every chunk
    fills its 4-KiB stage window to the last byte); `one_half` = every kept value of a token in channels 0..63 (value tiles of the upper half and key tiles of channels
    64..127 are EMPTY, the others carry twice the usual non-zeros); `zero_block` = 64 tokens of exact zeros in the middle of the cache (zero-length chunks)."""
    from mustafar_amd import _lib
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    lib = _lib.load()
    torch.manual_seed(31)
    bsz, D, L0 = 2, 128, 1024 + 40
    C, groups = 1024, hq // hkv
    s = 0.0 if kind == "dense" else 0.7
    K_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V_all = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    if kind == "one_half":
        K_all[..., 64:] *= 1e-3
        V_all[..., 64:] *= 1e-3
    if kind == "zero_block":
        K_all[:, :, 448:512] = 0
        V_all[:, :, 448:512] = 0
    Kp, Vp = K_all.clone(), V_all.clone()
    if s > 0:
        Kp[:, :, :C] = torch.from_numpy(orc.prune_magnitude(K_all[:, :, :C].cpu().numpy(), s)).to(DEV)
        Vp[:, :, :C] = torch.from_numpy(orc.prune_magnitude(V_all[:, :, :C].cpu().numpy(), s)).to(DEV)
    q, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    want = _dense(q, torch.cat([Kp, kn], 2), torch.cat([Vp, vn], 2), 0, 0.7, groups)
    outs = []
    try:
        for small, tbw, form in ((2, 0, 4), (0, 0, 3), (0, 2, 3)):        # small-launch kernel; super-block kernel, two blocks per workgroup; four
            assert lib.mustafar_tune(11, small) == 0 and lib.mustafar_tune(1, tbw) == 0
            a = MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, k_sparsity=s, v_sparsity=s, api="fused", arena=True,
                                                 structure="one_pass"))
            past = a.to_fused(a.build_cache(K_all.clone(), V_all.clone()))
            out, _ = a.decode(q, kn, vn, past)
            assert (lib.mustafar_last_decode_choice() >> 8) & 15 == form
            torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=2e-3, msg=lambda m, f=(small, tbw): f"form {f}: {m}")
            outs.append(out.float())
        scale = float(want.abs().max())
        for o in outs[1:]:
            assert float((o - outs[0]).abs().max()) <= 2 * 2.0 ** -11 * scale + 1e-4
    finally:
        lib.mustafar_tune(11, 1)
        lib.mustafar_tune(1, 0)
