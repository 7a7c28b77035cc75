"""GPU: growth of the compressed cache by extents (mustafar_amd/cache.py, mustafar_decode_attention_extents).

The 256-token trigger of models/llama_mustafar_kernel.py:324-398 appends to the compressed cache; the reference re-copies the cache
to do so.  With `MustafarConfig.extents` the trigger compresses its 256 tokens into an extent of their own and lists it in a device
table; the pair form of the one-pass launch reads the blocks behind the base tokens through that table.  Held here:
  * the same decode sequence through several triggers with extents and with the in-place append gives the SAME outputs (bit for
    bit: the same blocks, the same workgroups, the same slabs) and the same cache in the reference layout, and both equal dense
    fp32 attention over oracle-pruned K / V;
  * nothing a captured graph holds moves: base arrays, extent table and windows keep their addresses across a trigger, and a graph
    of the step BEHIND a trigger, captured before it, replays correctly after it;
  * a full table is answered by one consolidation, a launch form that cannot read extents by the in-place append.
"""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
R = 32


def _attn(hq, hkv, **kw):
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    return MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, k_sparsity=0.7, v_sparsity=0.7,
                                            api="fused", arena=True, **kw))


def _dense(q, K_all, V_all, C, s, groups):
    K, V = K_all.clone(), V_all.clone()
    K[:, :, :C] = torch.from_numpy(orc.prune_magnitude(K_all[:, :, :C].cpu().numpy(), s)).to(K.device)
    V[:, :, :C] = torch.from_numpy(orc.prune_magnitude(V_all[:, :, :C].cpu().numpy(), s)).to(V.device)
    Kr = K.double().repeat_interleave(groups, dim=1)
    Vr = V.double().repeat_interleave(groups, dim=1)
    sc = torch.matmul(q.double(), Kr.transpose(2, 3)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(sc, -1), Vr).float()


def _run(attn, K0, V0, qs, ks, vs):
    past = attn.to_fused(attn.build_cache(K0.clone(), V0.clone()))
    outs = []
    for q, k, v in zip(qs, ks, vs):
        out, past = attn.decode(q, k, v, past)
        outs.append(out)
    return outs, past


# (hq, hkv): GQA-4 on the three engines; MHA (G = 1) and GQA-2 (G = 2) on the v_fma_mix engine their pair form runs (round 4: before,
# only group counts % 4 == 0 could read extents)
@pytest.mark.parametrize("engine,hq,hkv", [("dot2", 8, 2), ("valu", 8, 2), ("mfma", 8, 2), (None, 8, 8), (None, 8, 4)])
def test_extents_equal_in_place_append_and_dense_through_three_triggers(engine, hq, hkv):
    torch.manual_seed(5)
    bsz, D = 2, 128
    L0, steps = 512 + R + 200, 56 + 2 * 256 + 9               # first trigger after 56 steps, then two more
    K0, V0 = (torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(2))
    qs = [torch.randn(bsz, hq, 1, D, device=DEV).half() for _ in range(steps)]
    ks = [torch.randn(bsz, hkv, 1, D, device=DEV).half() for _ in range(steps)]
    vs = [torch.randn(bsz, hkv, 1, D, device=DEV).half() for _ in range(steps)]
    o_ext, p_ext = _run(_attn(hq, hkv, engine=engine, extents=True), K0, V0, qs, ks, vs)
    o_inp, p_inp = _run(_attn(hq, hkv, engine=engine, extents=False), K0, V0, qs, ks, vs)
    assert len(p_ext[0].extents) == 3 and len(p_ext[2].extents) == 3 and not p_inp[0].extents
    assert p_ext[0].tokens == 512 and p_ext[0].total_tokens == p_ext[4] == p_inp[4] == p_inp[0].tokens == 1280
    for i, (a, b) in enumerate(zip(o_ext, o_inp)):
        assert torch.equal(a, b), f"step {i}: extents and in-place append differ"
    for side in (0, 2):                                        # the caches, in the reference layout
        a, b = p_ext[side].to_reference(), p_inp[side].to_reference()
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
        assert torch.equal(torch.cat(list(a[2])), torch.cat(list(b[2])))
    K_all = torch.cat([K0] + ks, 2)
    V_all = torch.cat([V0] + vs, 2)
    want = _dense(qs[-1], K_all, V_all, 1280, 0.7, hq // hkv)
    torch.testing.assert_close(o_ext[-1].float(), want, rtol=4e-3, atol=2e-3)
    # the extents cost what they hold: a few per cent over the bytes in use, nothing re-housed (+ the device table of their views:
    # 512 entries x 56 bytes, a constant that does not grow with the cache)
    table = p_ext[0].MAX_EXTENTS * p_ext[0].VIEW_BYTES
    assert p_ext[0].bytes_reserved() <= p_ext[0].bytes_in_use() * 1.06 + 4096 * p_ext[0].heads + table


def test_addresses_survive_a_trigger_and_a_graph_captured_ahead_replays_behind_it():
    from mustafar_amd import _lib
    lib = _lib.load()
    torch.manual_seed(6)
    bsz, hq, hkv, D = 2, 8, 2, 128
    L0 = 768 + R + 250                                         # the trigger fires at the 6th decode step
    K0, V0 = (torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(2))
    attn = _attn(hq, hkv)
    past = attn.to_fused(attn.build_cache(K0.clone(), V0.clone()))
    hist_k, hist_v = [K0], [V0]
    def new():
        return tuple(torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    for _ in range(3):
        q, k, v = new()
        _, past = attn.decode(q, k, v, past)
        hist_k.append(k); hist_v.append(v)
    sig = (past[0].signature(), past[2].signature(), past[1].buf.data_ptr(), past[3].buf.data_ptr())
    # the graph of the first step BEHIND the trigger, captured three steps before it fires: 256 more compressed tokens, windows at R rows
    import copy
    kw, vw = copy.copy(past[1]), copy.copy(past[3])
    kw.len = vw.len = R
    fut = (past[0], kw, past[2], vw, past[4] + 256, past[5] + 3)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    gq, gk, gv = new()
    attn.decode_fused(gq, gk, gv, (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5]))   # scratch outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        g_out, _ = attn.decode_fused(gq, gk, gv, fut, step_counter=counter)
    for _ in range(3):                                         # ... the trigger fires in the third of these
        q, k, v = new()
        _, past = attn.decode(q, k, v, past)
        hist_k.append(k); hist_v.append(v)
    assert past[4] == 1024 and len(past[0].extents) == 1
    assert sig == (past[0].signature(), past[2].signature(), past[1].buf.data_ptr(), past[3].buf.data_ptr()), "a trigger moved something a graph holds"
    counter.zero_()
    g.replay()                                                 # appends (gk, gv) to the windows and attends over 1024 + R + 1 tokens
    torch.cuda.synchronize()
    K_all = torch.cat(hist_k + [gk], 2)
    V_all = torch.cat(hist_v + [gv], 2)
    want = _dense(gq, K_all, V_all, 1024, 0.7, hq // hkv)
    torch.testing.assert_close(g_out.float(), want, rtol=4e-3, atol=2e-3)


def test_full_table_consolidates_and_other_launch_forms_append_in_place(monkeypatch):
    from mustafar_amd.cache import CompressedArena
    torch.manual_seed(7)
    bsz, hq, hkv, D = 1, 8, 2, 128
    L0, steps = 256 + R + 255, 1 + 3 * 256 + 5
    K0, V0 = (torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(2))
    qs = [torch.randn(bsz, hq, 1, D, device=DEV).half() for _ in range(steps)]
    ks = [torch.randn(bsz, hkv, 1, D, device=DEV).half() for _ in range(steps)]
    vs = [torch.randn(bsz, hkv, 1, D, device=DEV).half() for _ in range(steps)]
    monkeypatch.setattr(CompressedArena, "MAX_EXTENTS", 2)
    o_ext, p_ext = _run(_attn(hq, hkv, extents=True), K0, V0, qs, ks, vs)       # 4 triggers: 2 extents, a consolidation, 1 more...
    assert p_ext[4] == 256 + 4 * 256 and p_ext[0].total_tokens == p_ext[4] and len(p_ext[0].extents) <= 2 and p_ext[0].tokens >= 768
    o_two, p_two = _run(_attn(hq, hkv, extents=True, structure="two_launch"), K0, V0, qs, ks, vs)
    assert not p_two[0].extents and p_two[0].tokens == p_two[4] == p_ext[4]      # two launches cannot read extents: in-place append
    want = _dense(qs[-1], torch.cat([K0] + ks, 2), torch.cat([V0] + vs, 2), p_ext[4], 0.7, hq // hkv)
    torch.testing.assert_close(o_ext[-1].float(), want, rtol=4e-3, atol=2e-3)
    torch.testing.assert_close(o_two[-1].float(), want, rtol=4e-3, atol=2e-3)
    a, b = p_ext[0].to_reference(), p_two[0].to_reference()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(torch.cat(list(a[2])), torch.cat(list(b[2])))


@pytest.mark.parametrize("hq,hkv", [(8, 2), (8, 8)])
def test_consolidate_on_the_device_equals_the_host_round_trip(hq, hkv, monkeypatch):
    """consolidate() (round 5): base + extents re-housed into one base by two launches -- the extents found through the device table the
    decode launch reads, their offsets shifted by lengths read on the device -- equals the round-4 path through the reference layout on the
    host (MUSTAFAR_CONSOLIDATE=host) and the cache's own reference layout, bit for bit; and decodes alike."""
    torch.manual_seed(15)
    bsz, D = 2, 128
    L0, steps = 512 + R + 255, 1 + 2 * 256 + 2
    K0, V0 = (torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(2))
    qs = [torch.randn(bsz, hq, 1, D, device=DEV).half() for _ in range(steps)]
    ks = [torch.randn(bsz, hkv, 1, D, device=DEV).half() for _ in range(steps)]
    vs = [torch.randn(bsz, hkv, 1, D, device=DEV).half() for _ in range(steps)]
    attn = _attn(hq, hkv)
    _, past = _run(attn, K0, V0, qs, ks, vs)
    assert len(past[0].extents) == 3 and past[0].tokens == 512 and past[4] == 512 + 768
    for side in (0, 2):
        arena = past[side]
        dev_c = arena.consolidate()
        monkeypatch.setenv("MUSTAFAR_CONSOLIDATE", "host")
        host_c = arena.consolidate()
        monkeypatch.delenv("MUSTAFAR_CONSOLIDATE")
        assert not dev_c.extents and dev_c.tokens == host_c.tokens == past[4] and torch.equal(dev_c.used, host_c.used)
        a, b, c = dev_c.to_reference(), host_c.to_reference(), arena.to_reference()
        for x, y in ((a, b), (a, c)):
            assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[3], y[3])
            assert torch.equal(torch.cat(list(x[2])).view(torch.int16), torch.cat(list(y[2])).view(torch.int16))
    q, k, v = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    merged = (past[0].consolidate(), past[1].clone(), past[2].consolidate(), past[3].clone(), past[4], past[5])
    merged[0].ext_table, merged[2].ext_table
    o_m, _ = attn.decode(q, k, v, merged)
    o_e, _ = attn.decode(q, k, v, (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5]))
    assert torch.equal(o_m, o_e) or (o_m.float() - o_e.float()).abs().max() <= 2 ** -10 * o_e.float().abs().max()   # (other workgroup boundaries: other slab sums)


def test_engine_switch_to_a_form_without_extents_consolidates():
    """A cache that grew by extents handed to a launch form that reads one view (two launches asked for): decode_fused re-houses it
    once and goes on; outputs equal dense attention before and after."""
    torch.manual_seed(8)
    bsz, hq, hkv, D = 1, 8, 2, 128
    L0 = 256 + R + 250
    K0, V0 = (torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(2))
    a_ext, a_two = _attn(hq, hkv), _attn(hq, hkv, structure="two_launch")
    past = a_ext.to_fused(a_ext.build_cache(K0.clone(), V0.clone()))
    hk, hv = [K0], [V0]
    for i in range(12):                                        # the trigger fires at step 6; from step 9 on the other form decodes
        q, k, v = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
        hk.append(k); hv.append(v)
        out, past = (a_ext if i < 9 else a_two).decode(q, k, v, past)
        if i in (8, 9, 11):
            want = _dense(q, torch.cat(hk, 2), torch.cat(hv, 2), past[4], 0.7, hq // hkv)
            torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=2e-3, msg=lambda m, i=i: f"step {i}: {m}")
        if i == 8:
            assert len(past[0].extents) == 1 and past[0].tokens == 256
    assert not past[0].extents and past[0].tokens == past[4] == 512


@pytest.mark.parametrize("hq,hkv", [(8, 2), (8, 8), (8, 4)])
def test_one_graph_serves_the_cache_through_two_triggers_with_device_side_T(hq, hkv):
    """`t_device` / `t_capacity`: the launch is sized for a capacity and reads the compressed tokens in use from device memory.  ONE
    captured graph of the step is replayed across two 256-token triggers (run eagerly between replays: an extent each, 256 added
    to the device T, 256 taken off the window counter); every checked step equals dense attention.  GQA-4, MHA and GQA-2."""
    from mustafar_amd import _lib
    lib = _lib.load()
    torch.manual_seed(9)
    bsz, D = 2, 128
    L0 = 512 + R + 250                                         # first trigger at the 6th decode step, the second 256 steps later
    K0, V0 = (torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(2))
    attn = _attn(hq, hkv)
    past = attn.to_fused(attn.build_cache(K0.clone(), V0.clone()))
    C0, cap = past[4], past[4] + 512
    t_dev = torch.tensor([C0], dtype=torch.int32, device=DEV)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    q, k, v = (torch.zeros(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    warm = (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5])
    attn.decode_fused(q, k, v, warm, step_counter=counter, t_device=t_dev, t_capacity=cap)    # scratch for the capacity, outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, _ = attn.decode_fused(q, k, v, past, step_counter=counter, t_device=t_dev, t_capacity=cap)
        _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream().cuda_stream, counter.data_ptr(), 1), "counter")
    sig = (past[0].signature(), past[2].signature())
    hk, hv = [K0], [V0]
    state, since, triggers = past, 0, 0                        # host view of the cache: `state` advanced by `since` replays
    for step in range(6 + 256 + 12):
        qn, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
        hk.append(kn); hv.append(vn)
        kv_len = state[5] + since + 1
        if (kv_len - R - state[4]) % 256 == 0 and state[1].len + since + 1 >= 256:   # this step fires the trigger: eagerly
            got, state = attn.decode(qn, kn, vn, attn.advance(state, since))
            since, triggers = 0, triggers + 1
            t_dev.add_(256)
            counter.add_(1 - 256)                              # the eager step appended a row, the trigger slid the windows by 256
            C_step = state[4] - 256                            # (the step itself still ran over the old compressed length)
        else:
            q.copy_(qn); k.copy_(kn); v.copy_(vn)
            g.replay()
            got, since, C_step = out, since + 1, state[4]
        if step in (0, 4, 5, 6, 7, 150, 261, 262, 263, 273):
            want = _dense(qn, torch.cat(hk, 2), torch.cat(hv, 2), C_step, 0.7, hq // hkv)
            torch.testing.assert_close(got.float(), want, rtol=4e-3, atol=2e-3, msg=lambda m, step=step: f"step {step}: {m}")
    assert triggers == 2 and state[4] == C0 + 512 == cap and len(state[0].extents) == 2
    assert sig == (state[0].signature(), state[2].signature())


def test_left_padding_mask_across_a_trigger_with_extents():
    """The hook's additive mask (models/llama_mustafar_kernel.py:293-301) over a cache that has grown by an extent: the columns of
    the extent's tokens and of the window (behind ALL compressed tokens) line up with the mask; dense fp32 attention over the
    unmasked columns is the reference."""
    NEG = torch.finfo(torch.float16).min
    torch.manual_seed(10)
    bsz, hq, hkv, D = 3, 8, 2, 128
    L0 = 256 + R + 252                                         # the trigger fires at the 4th decode step
    pads = (0, 300, 530)                                       # row 1: inside the base, row 2: base and part of the extent
    K0, V0 = (torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(2))
    attn = _attn(hq, hkv)
    past = attn.to_fused(attn.build_cache(K0.clone(), V0.clone()))
    hk, hv = [K0], [V0]
    for step in range(9):
        qn, kn, vn = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
        hk.append(kn); hv.append(vn)
        kv_len = past[5] + 1
        mask = torch.zeros((bsz, 1, 1, kv_len), dtype=torch.float16, device=DEV)
        for b, p in enumerate(pads):
            mask[b, :, :, :p] = NEG
        C_step = past[4]
        out, past = attn.decode(qn, kn, vn, past, attention_mask=mask)
        K_all, V_all = torch.cat(hk, 2), torch.cat(hv, 2)
        K, V = K_all.clone(), V_all.clone()
        K[:, :, :C_step] = torch.from_numpy(orc.prune_magnitude(K_all[:, :, :C_step].cpu().numpy(), 0.7)).to(DEV)
        V[:, :, :C_step] = torch.from_numpy(orc.prune_magnitude(V_all[:, :, :C_step].cpu().numpy(), 0.7)).to(DEV)
        s = torch.matmul(qn.float(), K.float().repeat_interleave(hq // hkv, dim=1).transpose(2, 3)) / math.sqrt(D)
        s = s.masked_fill(mask < 0, float("-inf"))
        want = torch.matmul(torch.softmax(s, -1), V.float().repeat_interleave(hq // hkv, dim=1))
        torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=2e-3, msg=lambda m, step=step: f"step {step}: {m}")
    assert past[4] == 512 and len(past[0].extents) == 1


@pytest.mark.parametrize("hq,hkv,prepared", [(8, 2, True), (8, 2, False), (8, 8, True)])
def test_batched_trigger_of_all_layers_equals_the_layer_by_layer_trigger(hq, hkv, prepared):
    """`run_triggers` (round 4): the trigger of every layer in two library calls and ONE host read, into a pooled allocation made ahead
    (`prepare_triggers`).  Three layers with different data go through two triggers: every step's output, the caches in the
    reference layout and the windows equal the layer-by-layer trigger that decode_fused runs itself, bit for bit."""
    torch.manual_seed(12)
    layers, bsz, D = 3, 2, 128
    L0, steps = 256 + R + 250, 6 + 256 + 3
    K0 = [torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(layers)]
    V0 = [torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(layers)]
    attn = _attn(hq, hkv)
    ref_p = [attn.to_fused(attn.build_cache(K0[l].clone(), V0[l].clone())) for l in range(layers)]
    bat_p = [attn.to_fused(attn.build_cache(K0[l].clone(), V0[l].clone())) for l in range(layers)]
    fired = 0
    for step in range(steps):
        pool = None
        qkv = [tuple(torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv)) for _ in range(layers)]
        if prepared and (bat_p[0][5] + 1 - R - bat_p[0][4]) % 256 == 0 and bat_p[0][1].len + 1 >= 256:   # this step will reach the trigger
            pool = attn.prepare_triggers(bat_p)
            assert pool is not None
        for l in range(layers):
            o_ref, ref_p[l] = attn.decode(*qkv[l], ref_p[l])
            o_bat, bat_p[l] = attn.decode_fused(*qkv[l], bat_p[l], defer_trigger=True)
            assert torch.equal(o_ref, o_bat), f"step {step} layer {l}"
        if attn.trigger_due(bat_p[0]):
            fired += 1
            assert all(attn.trigger_due(p) for p in bat_p)
            bat_p = attn.run_triggers(bat_p, pool)
        for l in range(layers):
            assert bat_p[l][4] == ref_p[l][4] and bat_p[l][1].len == ref_p[l][1].len == bat_p[l][3].len
    assert fired == 2 and all(len(p[0].extents) == 2 and p[0].total_tokens == 256 + 512 for p in bat_p)
    for l in range(layers):
        for side in (0, 2):
            a, b = bat_p[l][side].to_reference(), ref_p[l][side].to_reference()
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
            assert torch.equal(torch.cat(list(a[2])), torch.cat(list(b[2])))
        assert torch.equal(bat_p[l][1].view(), ref_p[l][1].view()) and torch.equal(bat_p[l][3].view(), ref_p[l][3].view())


def test_batched_trigger_redoes_a_layer_whose_rows_are_full_of_ties():
    """A layer whose 256 window rows keep EVERY value (all magnitudes equal: ties at the threshold are kept, model :107) outgrows the
    pooled extent's region: its flag comes back set, it is redone on its own at the measured size (its raw rows are still in place:
    nothing slides before every flag has been seen), the other layers are untouched -- and all equal the layer-by-layer trigger."""
    torch.manual_seed(13)
    layers, bsz, hq, hkv, D = 3, 1, 8, 2, 128
    L0 = 256 + R + 255                                   # the first decode step reaches the trigger
    K0 = [torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(layers)]
    V0 = [torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(layers)]
    sign = torch.where(torch.rand(bsz, hkv, 256, D, device=DEV) < 0.5, -1.0, 1.0).half()
    K0[1][:, :, 256:512] = 0.5 * sign                    # layer 1: the rows of the coming trigger tie everywhere (K only)
    attn = _attn(hq, hkv)
    ref_p = [attn.to_fused(attn.build_cache(K0[l].clone(), V0[l].clone())) for l in range(layers)]
    bat_p = [attn.to_fused(attn.build_cache(K0[l].clone(), V0[l].clone())) for l in range(layers)]
    qkv = [tuple(torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv)) for _ in range(layers)]
    for l in range(layers):
        _, ref_p[l] = attn.decode(*qkv[l], ref_p[l])
        _, bat_p[l] = attn.decode_fused(*qkv[l], bat_p[l], defer_trigger=True)
    bat_p = attn.run_triggers(bat_p, attn.prepare_triggers(bat_p))
    for l in range(layers):
        assert bat_p[l][4] == ref_p[l][4] == 512 and len(bat_p[l][0].extents) == 1 and bat_p[l][1].len == ref_p[l][1].len == R
        for side in (0, 2):
            a, b = bat_p[l][side].to_reference(), ref_p[l][side].to_reference()
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(torch.cat(list(a[2])), torch.cat(list(b[2])))
        assert torch.equal(bat_p[l][1].view(), ref_p[l][1].view())
    assert int(bat_p[1][0].extents[0].used.max()) == 256 * 128       # every value of the tied rows was kept
    q2 = [tuple(torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv)) for _ in range(layers)]
    for l in range(layers):                               # and the caches decode alike afterwards
        o_ref, _ = attn.decode(*q2[l], ref_p[l])
        o_bat, _ = attn.decode(*q2[l], bat_p[l])
        assert torch.equal(o_ref, o_bat)


@pytest.mark.parametrize("residual", [128, 300])
def test_batched_trigger_with_a_long_residual_window(residual):
    """residual_length is the reference's config value (model :53, mem_spd_test.py:9 sets 32): with 128 or 300 rows staying behind a
    trigger the slide moves more than the 64 rows the round-4 kernel was limited to -- at 300 the ranges overlap (more rows stay than
    leave).  The batched trigger (run_triggers) and the layer-by-layer one (decode) must agree bit for bit, and both with dense
    attention over oracle-pruned K / V."""
    torch.manual_seed(14)
    layers, bsz, hq, hkv, D = 2, 1, 8, 2, 128
    attn = _attn(hq, hkv, residual_length=residual)
    L0 = 256 + residual + 254                              # the second decode step reaches the trigger
    K0 = [torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(layers)]
    V0 = [torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(layers)]
    ref_p = [attn.to_fused(attn.build_cache(K0[l].clone(), V0[l].clone())) for l in range(layers)]
    bat_p = [attn.to_fused(attn.build_cache(K0[l].clone(), V0[l].clone())) for l in range(layers)]
    Kall, Vall = [k.clone() for k in K0], [v.clone() for v in V0]
    fired = 0
    for step in range(4):
        qkv = [tuple(torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv)) for _ in range(layers)]
        for l in range(layers):
            Kall[l] = torch.cat([Kall[l], qkv[l][1]], 2)
            Vall[l] = torch.cat([Vall[l], qkv[l][2]], 2)
            C = ref_p[l][4]
            o_ref, ref_p[l] = attn.decode(*qkv[l], ref_p[l])
            o_bat, bat_p[l] = attn.decode_fused(*qkv[l], bat_p[l], defer_trigger=True)
            assert torch.equal(o_ref, o_bat), f"step {step} layer {l}"
            want = _dense(qkv[l][0], Kall[l], Vall[l], C, 0.7, hq // hkv)
            assert torch.allclose(o_bat.float(), want, rtol=4e-3, atol=2e-3)
        if attn.trigger_due(bat_p[0]):
            fired += 1
            bat_p = attn.run_triggers(bat_p, attn.prepare_triggers(bat_p))
        for l in range(layers):
            assert bat_p[l][4] == ref_p[l][4] and bat_p[l][1].len == ref_p[l][1].len == bat_p[l][3].len
            assert torch.equal(bat_p[l][1].view(), ref_p[l][1].view()) and torch.equal(bat_p[l][3].view(), ref_p[l][3].view())
    assert fired == 1 and bat_p[0][4] == 512 and bat_p[0][1].len == residual + 2


@pytest.mark.parametrize("batched", [False, True])
def test_a_timed_out_one_pass_compression_is_repeated_in_the_two_pass_form(batched):
    """Flag bit 1 of the one-pass compression launch -- a block gave up waiting for the lengths of the blocks in front of it; it relies on
    in-order dispatch and has never been seen -- is FORCED here (mustafar_compress_test_skip_publish: block 1 of the next launch keeps its
    length to itself, blocks 2 and 3 of every head time out): round 4 raised ArenaAppendTimeout, round 5 repeats the append once in the
    two-pass form (the raw rows are still in place, the call is idempotent) and counts it.  Outputs, caches and windows equal the
    undisturbed run bit for bit, layer by layer (decode) and through the batched trigger of all layers (run_triggers)."""
    from mustafar_amd import _lib, cache
    torch.manual_seed(16)
    layers, bsz, hq, hkv, D = 2, 1, 8, 2, 128
    L0 = 256 + R + 255                                   # the first decode step reaches the trigger
    K0 = [torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(layers)]
    V0 = [torch.randn(bsz, hkv, L0, D, device=DEV).half() for _ in range(layers)]
    attn = _attn(hq, hkv)
    ref_p = [attn.to_fused(attn.build_cache(K0[l].clone(), V0[l].clone())) for l in range(layers)]
    tst_p = [attn.to_fused(attn.build_cache(K0[l].clone(), V0[l].clone())) for l in range(layers)]
    qkv = [tuple(torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv)) for _ in range(layers)]
    for l in range(layers):
        _, ref_p[l] = attn.decode(*qkv[l], ref_p[l])
    before = cache.compress_fallbacks
    lib = _lib.load()
    if batched:
        for l in range(layers):
            _, tst_p[l] = attn.decode_fused(*qkv[l], tst_p[l], defer_trigger=True)
        pool = attn.prepare_triggers(tst_p)
        assert lib.mustafar_compress_test_skip_publish(1) == 0
        tst_p = attn.run_triggers(tst_p, pool)
    else:
        for l in range(layers):
            if l == 0:
                assert lib.mustafar_compress_test_skip_publish(1) == 0
            _, tst_p[l] = attn.decode(*qkv[l], tst_p[l])
    assert cache.compress_fallbacks == before + 1, "the forced time-out was not taken (or taken more than once)"
    for l in range(layers):
        assert tst_p[l][4] == ref_p[l][4] == 512 and len(tst_p[l][0].extents) == 1 and tst_p[l][1].len == ref_p[l][1].len == R
        for side in (0, 2):
            a, b = tst_p[l][side].to_reference(), ref_p[l][side].to_reference()
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(torch.cat(list(a[2])).view(torch.int16), torch.cat(list(b[2])).view(torch.int16))
        assert torch.equal(tst_p[l][1].view(), ref_p[l][1].view())
    q2 = [tuple(torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv)) for _ in range(layers)]
    for l in range(layers):
        o_ref, _ = attn.decode(*q2[l], ref_p[l])
        o_tst, _ = attn.decode(*q2[l], tst_p[l])
        assert torch.equal(o_ref, o_tst)


def test_a_device_side_T_needs_the_step_counter():
    """`t_device` sizes the launch for a capacity and is meant for captured graphs: an eager call (no `step_counter`) is refused with a
    clear error instead of launching at the capacity (and contradicting the eager mask-length check)."""
    torch.manual_seed(14)
    bsz, hq, hkv, D = 1, 8, 2, 128
    K0, V0 = (torch.randn(bsz, hkv, 256 + R + 10, D, device=DEV).half() for _ in range(2))
    attn = _attn(hq, hkv)
    past = attn.to_fused(attn.build_cache(K0, V0))
    q, k, v = (torch.randn(bsz, h, 1, D, device=DEV).half() for h in (hq, hkv, hkv))
    t_dev = torch.tensor([256], dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError, match="step_counter"):
        attn.decode_fused(q, k, v, past, t_device=t_dev, t_capacity=512)
