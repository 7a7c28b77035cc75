"""GPU: the path bench.py TIMES, at the shapes it times it -- fused decode over the arena cache, eagerly and as a
replayed hipGraph, at the BASELINE configs c2-c5 (one layer each; bench.py runs 32 of them).  At these sizes the launches
carry what no small test has: 256 rows, Split_K ~ 32, window workgroups riding both SpMV grids, both key launch forms.

s4 / s32 = c3's geometry and batch at L = 4096 / 32768: the ends of the axis the metric is quoted on (BASELINE.json.metric:
Llama-3-8B, 70 %, seq_len 4k-32k; bench.py's `seq_sweep`).

Checks, per config:
  * fused (eager) == dense fp32 attention over K/V pruned by the prune rule (the GPU prune kernel, itself held bit-exact
    to the reference's dh_prune_* fixtures and, here, to the CPU oracle on one head at full T), within `DENSE_ULPS` fp16 ulp of
    the OUTPUT SCALE (+ 1e-4): a bound that sees one lost 64-token block at every config (round 3 used rtol 4e-3, atol 2e-3: an
    absolute term near the size of the outputs themselves at c4 / c5; the negative control below holds the new bar to that)
  * fused == the two reference entry points with PyTorch glue (api="native") within 2 fp16 ulp of the output scale: eager steps,
    every replay of the captured graph, and the steps around the compression trigger
  * the same step replayed from a captured graph >= 3 times while the window grows
  * c3, c5: eager steps across the 256-token compression trigger (32nd decode step), arena append included
  * c4: NEGATIVE CONTROL -- one 64-token block of one kv-head's values dropped from the cache: both comparators must fail
  * c3/c4/c5/s4/s32: the C oracle's key and value SpMV on ONE kv-head at full T against the HIP result (fp16_bound, tests/util.py)
"""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.util import DENSE_ULPS, NATIVE_ULPS, excess, fp16_bound

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = {  # BASELINE.json configs[1..4]: (Hq, Hkv, sparsity, L, batch)
    "c2": (32, 32, 0.7, 4096, 1),
    "c3": (32, 8, 0.7, 8192, 8),
    "c4": (32, 8, 0.8, 32768, 4),
    "c5": (32, 8, 0.7, 16384, 16),
    "g2": (32, 16, 0.7, 4096, 4),      # GQA-2 at size: the G = 2 instantiation of the pair kernel (round 4)
    "m8": (32, 32, 0.7, 8192, 8),      # MHA at size (round 6): 32 / 32 heads, L = 8192, batch 8 -- four blocks per workgroup on the G = 1 form, the shape
                                       # round 5's first super-block build got wrong (block B's e segments read at G = 4's offset) while every suite case passed
    "s4": (32, 8, 0.7, 4096, 8),
    "s32": (32, 8, 0.7, 32768, 8),
    "c3w": (32, 8, 0.7, 7936 + 32 + 255, 8),   # c3's geometry with a window one token short of the trigger (round 5: off the grid of whole rounds)
}



class DenseRef:
    """Dense attention over pruned-but-dense K/V (the relation between the reference's kernel model and its dense model,
    llama_mustafar_Kt_Mag_Vt_Mag.py:873, :963, :974), kept on the GPU in fp16 and evaluated in fp32 per kv-head group."""

    def __init__(self, K, V, C, s, groups):
        from mustafar_amd import compression
        self.prune = lambda x: compression.prune_magnitude(x.contiguous(), s)
        self.K, self.V, self.C, self.groups = K.clone(), V.clone(), C, groups
        self.K[:, :, :C] = self.prune(K[:, :, :C])
        self.V[:, :, :C] = self.prune(V[:, :, :C])

    def append(self, k, v):
        self.K, self.V = torch.cat([self.K, k], 2), torch.cat([self.V, v], 2)

    def compress_next_256(self):
        C = self.C
        self.K[:, :, C:C + 256] = self.prune(self.K[:, :, C:C + 256])
        self.V[:, :, C:C + 256] = self.prune(self.V[:, :, C:C + 256])
        self.C += 256

    def __call__(self, q):
        B, Hq, _, D = q.shape
        Hkv = self.K.shape[1]
        out = torch.empty((B, Hq, 1, D), dtype=torch.float32, device=q.device)
        qg = q.view(B, Hkv, self.groups, D).float()
        for b in range(B):     # one batch entry at a time bounds the fp32 temporaries (c4: 32 k tokens)
            s = torch.einsum("hgd,htd->hgt", qg[b], self.K[b].float()) / math.sqrt(D)
            out[b] = torch.einsum("hgt,htd->hgd", torch.softmax(s, -1), self.V[b].float()).reshape(Hq, 1, D)
        return out


def _setup(name, arena=True):
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    Hq, Hkv, s, L, batch = CASES[name]
    torch.manual_seed(42)
    cfg = MustafarConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, k_sparsity=s, v_sparsity=s, residual_length=32,
                         api="fused", arena=arena)
    attn = MustafarAttention(cfg)
    K = torch.randn(batch, Hkv, L, 128, device=DEV).half()
    V = torch.randn(batch, Hkv, L, 128, device=DEV).half()
    T = ((L - 32) // 256) * 256
    past = attn.to_fused(attn.build_cache(K, V))
    assert past[4] == T
    ref = DenseRef(K, V, T, s, Hq // Hkv)
    return attn, cfg, past, ref, (Hq, Hkv, batch, T)


def _new(batch, Hq, Hkv):
    return (torch.randn(batch, Hq, 1, 128, device=DEV).half(), torch.randn(batch, Hkv, 1, 128, device=DEV).half(),
            torch.randn(batch, Hkv, 1, 128, device=DEV).half())


@pytest.mark.parametrize("name,structure", [("c2", 2), ("c3", 2), ("c4", 2), ("c5", 2), ("s4", 2), ("s32", 2), ("g2", 2), ("m8", 2), ("c3", 0), ("c5", 1)])
def test_fused_arena_eager_and_graph_at_bench_shape(name, structure):
    """structure 2 = the library's own choice by size (the one-pass launch at every config since round 3); the two-launch
    structure is forced once at c3, the one-pass launch named explicitly once at c5."""
    from mustafar_amd import _lib
    assert _lib.load().mustafar_set_onepass(structure) == 0
    try:
        _fused_arena_eager_and_graph(name)
    finally:
        _lib.load().mustafar_set_onepass(2)


def _fused_arena_eager_and_graph(name):
    from mustafar_amd import _lib
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    attn, cfg, past, ref, (Hq, Hkv, batch, T) = _setup(name)
    from mustafar_amd.cache import CompressedArena
    assert isinstance(past[0], CompressedArena) and isinstance(past[2], CompressedArena)
    # the same cache in the reference layout for the unfused call sequence
    native = MustafarAttention(MustafarConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, k_sparsity=cfg.k_sparsity,
                                              v_sparsity=cfg.v_sparsity, residual_length=32, api="native"))
    past_n = (past[0].to_reference(), past[1].view().clone(), past[2].to_reference(), past[3].view().clone(), past[4], past[5])
    # ---- eager, 3 steps
    for _ in range(3):
        q, k, v = _new(batch, Hq, Hkv)
        ref.append(k, v)
        out, past = attn.decode(q, k, v, past)
        out_n, past_n = native.decode(q, k, v, past_n)
        assert excess(out, ref(q), DENSE_ULPS) <= 1.0, "fused vs dense attention over the pruned K / V"
        assert excess(out, out_n, NATIVE_ULPS) <= 1.0, "fused and unfused call sequences disagree beyond 2 ulp of the output scale"
    # ---- the same call captured once and replayed (bench.py's timed form)
    q, k, v = (torch.zeros_like(t) for t in _new(batch, Hq, Hkv))
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    lib = _lib.load()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, _ = attn.decode_fused(q, k, v, past, step_counter=counter)
        _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream().cuda_stream, counter.data_ptr(), 1), "counter")
    for _ in range(4):
        qn, kn, vn = _new(batch, Hq, Hkv)
        q.copy_(qn); k.copy_(kn); v.copy_(vn)
        ref.append(kn, vn)
        g.replay()
        out_n, past_n = native.decode(qn, kn, vn, past_n)
        assert excess(out, ref(qn), DENSE_ULPS) <= 1.0, "replayed step vs dense attention"
        assert excess(out, out_n, NATIVE_ULPS) <= 1.0, "replayed step vs the unfused call sequence"
    del past_n
    past = attn.advance(past, 4)
    assert past[1].len == past[5] - T
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["c3", "c5"])
def test_fused_arena_across_the_compression_trigger_at_bench_shape(name):
    """L = 8192 / 16384 leave a 256-token window: the 32nd decode step fires the trigger (model :324): prune + compression of 256
    tokens per head into an extent of the arena, then decode continues over T + 256 compressed tokens."""
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    attn, cfg, past, ref, (Hq, Hkv, batch, T) = _setup(name)
    native = MustafarAttention(MustafarConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, k_sparsity=cfg.k_sparsity,
                                              v_sparsity=cfg.v_sparsity, residual_length=32, api="native"))
    past_n = (past[0].to_reference(), past[1].view().clone(), past[2].to_reference(), past[3].view().clone(), past[4], past[5])
    fired = 0
    for step in range(35):
        q, k, v = _new(batch, Hq, Hkv)
        ref.append(k, v)
        C_before = past[4]
        out, past = attn.decode(q, k, v, past)
        out_n, past_n = native.decode(q, k, v, past_n)      # (the unfused sequence fires its own trigger at the same step, model :324-398)
        if step in (0, 30, 31, 32, 34):
            assert excess(out, ref(q), DENSE_ULPS) <= 1.0, f"step {step}: fused vs dense attention"
            assert excess(out, out_n, NATIVE_ULPS) <= 1.0, f"step {step}: fused vs the unfused call sequence"
        if past[4] != C_before:
            fired += 1
            ref.compress_next_256()
            # (the 256 tokens arrive as an extent behind the T base tokens, nothing re-housed: tests/test_gpu_extents.py)
            assert step == 31 and past[4] == T + 256 and past[0].total_tokens == T + 256 and past[0].tokens == T and past[1].len == 32
    assert fired == 1
    torch.cuda.empty_cache()


def test_c3_two_triggers_past_the_resident_round_at_T_8448():
    """Every BASELINE point sits at T = 2^k - 256, just under ONE resident round of workgroups (c3: 1984 of the chip's 2048 slots); a
    cache that has grown by two 256-token extents does not (T = 8448: 2112 workgroups of four blocks, 33 per head).  c3's geometry through
    two triggers -- 7936 -> 8192 -> 8448 compressed tokens, the last 512 in two extents: the fused entry point, eager and as a replayed
    graph, against dense attention over the pruned K / V and against the unfused call sequence at each length."""
    from mustafar_amd import _lib
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    attn, cfg, past, ref, (Hq, Hkv, batch, T) = _setup("c3w")
    assert T == 7936 and past[1].len == 32 + 255   # (the window holds the residual 32 rows too: the first decode step reaches the trigger)
    native = MustafarAttention(MustafarConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, k_sparsity=cfg.k_sparsity,
                                              v_sparsity=cfg.v_sparsity, residual_length=32, api="native"))
    pend_k, pend_v = [], []

    def flush():                      # (the dense reference's K / V grow by one torch.cat per check, not per step)
        if pend_k:
            ref.append(torch.cat(pend_k, 2), torch.cat(pend_v, 2))
            pend_k.clear(); pend_v.clear()

    fired, checks = 0, 0
    check_at = (0, 1, 2, 255, 256, 257, 259)
    for step in range(1 + 256 + 3):
        q, k, v = _new(batch, Hq, Hkv)
        pend_k.append(k); pend_v.append(v)
        C_before = past[4]
        past_n = None
        if step in check_at:    # the same cache in the reference layout, taken BEFORE the step (the unfused sequence runs its own trigger, model :324-398)
            past_n = (past[0].to_reference(), past[1].view().clone(), past[2].to_reference(), past[3].view().clone(), past[4], past[5])
        out, past = attn.decode(q, k, v, past)
        if step in check_at:
            flush()
            checks += 1
            assert excess(out, ref(q), DENSE_ULPS) <= 1.0, f"step {step} (T = {C_before}): fused vs dense attention"
            out_n, _ = native.decode(q, k, v, past_n)
            assert excess(out, out_n, NATIVE_ULPS) <= 1.0, f"step {step}: fused vs the unfused call sequence"
            del past_n
        if past[4] != C_before:
            fired += 1
            flush()
            ref.compress_next_256()
            assert past[0].tokens == T and len(past[0].extents) == fired and past[0].total_tokens == T + 256 * fired
    assert fired == 2 and past[4] == 8448 and checks == 7
    # ---- the step at T = 8448 captured once and replayed
    q, k, v = (torch.zeros_like(t) for t in _new(batch, Hq, Hkv))
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    lib = _lib.load()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, _ = attn.decode_fused(q, k, v, past, step_counter=counter)
        _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream().cuda_stream, counter.data_ptr(), 1), "counter")
    for _ in range(3):
        qn, kn, vn = _new(batch, Hq, Hkv)
        q.copy_(qn); k.copy_(kn); v.copy_(vn)
        ref.append(kn, vn)
        g.replay()
        assert excess(out, ref(qn), DENSE_ULPS) <= 1.0, "replayed step at T = 8448 vs dense attention"
    torch.cuda.empty_cache()


def test_a_dropped_block_fails_the_comparators_at_c4():
    """NEGATIVE CONTROL.  c4 has the longest rows (32 512 compressed tokens): one 64-token block is 0.2 % of a row, and round 3's
    absolute tolerance (2e-3 on outputs of ~0.02) could not see it go missing.  Drop one block of one kv-head's VALUES from the
    cache (its 128 bitmaps cleared: the tiles contribute nothing, every other tile keeps its stream position) and both comparators
    of this file must fail on the rows of that head, and pass on every other row."""
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    attn, cfg, past, ref, (Hq, Hkv, batch, T) = _setup("c4")
    native = MustafarAttention(MustafarConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, k_sparsity=cfg.k_sparsity,
                                              v_sparsity=cfg.v_sparsity, residual_length=32, api="native"))
    past_n = (past[0].to_reference(), past[1].view().clone(), past[2].to_reference(), past[3].view().clone(), past[4], past[5])
    q, k, v = _new(batch, Hq, Hkv)
    ref.append(k, v)
    want, (out_n, _) = ref(q), native.decode(q, k, v, past_n)
    # (queries, keys and values stay random and the block is an ordinary one: nothing is rigged towards it)
    h, blk = 5, 200                                      # kv-head 5 of batch entry 0, tokens 12 800 .. 12 863
    saved = past[2].bmp[h, blk * 128:(blk + 1) * 128].clone()

    def fused_step():                                    # private windows: the step can be taken twice
        return attn.decode(q, k, v, (past[0], past[1].clone(), past[2], past[3].clone(), past[4], past[5]))[0]

    out_ok = fused_step()
    assert excess(out_ok, want, DENSE_ULPS) <= 1.0 and excess(out_ok, out_n, NATIVE_ULPS) <= 1.0      # intact cache: both pass
    past[2].bmp[h, blk * 128:(blk + 1) * 128] = 0
    out_bad = fused_step()
    rows = slice(h // Hkv, h // Hkv + 1), slice((h % Hkv) * (Hq // Hkv), (h % Hkv + 1) * (Hq // Hkv))
    assert excess(out_bad[rows], want[rows], DENSE_ULPS) > 1.0, "the dense comparator does not see a dropped 64-token block at c4"
    assert excess(out_bad[rows], out_n[rows], NATIVE_ULPS) > 1.0, "the unfused comparator does not see a dropped 64-token block at c4"
    others = torch.ones(batch, Hq, dtype=torch.bool, device=DEV)
    others[rows] = False
    assert excess(out_bad[others], want[others], DENSE_ULPS) <= 1.0                                  # every other row is untouched
    past[2].bmp[h, blk * 128:(blk + 1) * 128] = saved
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["c3", "c4", "c5", "s4", "s32", "m8"])
def test_one_head_at_full_length_against_the_c_oracle(name):
    """The CPU oracle finishes one kv-head at full T in seconds: prune, compress and both SpMVs of that head, HIP vs C."""
    from mustafar_amd import compression, mustafar_package as mp
    Hq, Hkv, s, L, batch = CASES[name]
    T, groups, D = ((L - 32) // 256) * 256, Hq // Hkv, 128
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, T, D)).astype(np.float16)
    xg = torch.from_numpy(x).to(DEV)
    want_pruned = orc.prune_magnitude(x, s)
    pruned = compression.prune_magnitude(xg, s)
    assert np.array_equal(pruned.cpu().numpy().view(np.uint16), want_pruned.view(np.uint16))
    q = rng.standard_normal((groups, 1, D)).astype(np.float16)
    p = torch.softmax(torch.from_numpy(rng.standard_normal((groups, 1, T)).astype(np.float32) * 2), -1).half().numpy()
    dense = want_pruned[0].astype(np.float64)
    for which in ("key", "value"):
        conv_g = compression.convert_key_batched if which == "key" else compression.convert_value_batched
        conv_o = orc.convert_key_batched if which == "key" else orc.convert_value_batched
        bmp, idx, nzs = conv_g(pruned)
        obmp, oidx, onzs = conv_o(want_pruned)
        assert np.array_equal(bmp.cpu().numpy(), obmp) and np.array_equal(idx.cpu().numpy(), oidx)
        assert np.array_equal(torch.cat(nzs).cpu().numpy().view(np.uint16), np.concatenate(onzs).view(np.uint16))
        off = torch.zeros(1, dtype=torch.int32, device=DEV)
        ooff = orc.nz_offset_from_idx(oidx)
        if which == "key":
            got = mp.mustafar_key_formulation(bmp, torch.cat(nzs), idx, off, torch.from_numpy(q).to(DEV), T, D, groups, groups)
            _, ref64 = orc.key_spmv(obmp, np.concatenate(onzs), oidx, ooff, q, T, D, groups, groups)
            sumabs = np.stack([np.abs(dense) @ np.abs(q[b, 0].astype(np.float64)) for b in range(groups)])[:, None]
        else:
            ws = torch.zeros(1, dtype=torch.float16, device=DEV)
            got = mp.mustafar_value_formulation(bmp, torch.cat(nzs), idx, off, torch.from_numpy(p).to(DEV), ws, D, T, groups, groups)
            _, ref64 = orc.value_spmv(obmp, np.concatenate(onzs), oidx, ooff, p, D, T, groups, groups)
            sumabs = np.stack([np.abs(p[b, 0].astype(np.float64)) @ np.abs(dense) for b in range(groups)])[:, None]
        err = np.abs(got.float().cpu().numpy().astype(np.float64) - ref64)
        assert (err <= fp16_bound(ref64, sumabs)).all(), f"{name} {which}: HIP SpMV outside the fp16 bound of the oracle's exact sum"


@pytest.mark.parametrize("name", ["c3", "c4", "c5"])
def test_one_pass_compression_is_bit_exact_at_bench_size(name):
    """`from_raw_pair` (compress_block_kernel: prune + compress + pack in one read, 124 - 508 blocks per head finding their stream
    positions from each other's published lengths) at the FULL size bench.py builds its caches at, against
      * the two-pass form on the same device data (prune_magnitude + convert_*_batched: tile_meta -> block_scan -> tile_pack),
        compared on the GPU: bitmaps, offsets, every head's stream and length;
      * the C oracle on ONE kv-head (prune + bitmaps + offsets + stream), bit for bit.
    (Before round 3 the one-pass form was held bit-exact only up to 1024 tokens x 8 heads.)"""
    from mustafar_amd import compression
    from mustafar_amd.cache import CompressedArena
    Hq, Hkv, s, L, batch = CASES[name]
    T = ((L - 32) // 256) * 256
    kth = compression.kth_from_sparsity(s, 128)
    g = torch.Generator(device=DEV).manual_seed(77)
    K = torch.randn((batch, Hkv, L, 128), device=DEV, generator=g).half()
    V = torch.randn((batch, Hkv, L, 128), device=DEV, generator=g).half()
    K[0, 0, 5, :] = 0.5                       # a row of ties (all 128 kept), a row of zeros, signed zeros: the rule's corners at full size
    K[-1, -1, T - 1, :] = 0
    V[0, 0, 7, ::2] = -0.0
    ka, va = CompressedArena.from_raw_pair(K, V, T, kth, kth)
    assert ka.tokens == va.tokens == T
    heads = batch * Hkv
    for which, arena, X in (("key", ka, K), ("value", va, V)):
        pruned = compression.prune_magnitude(X[:, :, :T].reshape(heads, T, 128).contiguous(), s)
        conv = compression.convert_key_batched if which == "key" else compression.convert_value_batched
        bmp, idx, nzs = conv(pruned)
        t2 = 2 * T
        assert torch.equal(arena.bmp[:, :t2], bmp.view(heads, t2)), f"{name} {which}: bitmaps differ from the two-pass form"
        assert torch.equal(arena.idx[:, :t2 + 1], idx.view(heads, t2 + 1)), f"{name} {which}: offsets differ from the two-pass form"
        used = arena.used
        assert [int(u) for u in used] == [n.numel() for n in nzs]
        for h in range(heads):
            assert torch.equal(arena.nz[h, :int(used[h])].view(torch.int16), nzs[h].view(torch.int16)), f"{name} {which}: stream of head {h}"
        # one kv-head against the C oracle (the head that carries the planted corner rows)
        h = 0
        x_h = X[0, 0, :T].cpu().numpy()[None]
        conv_o = orc.convert_key_batched if which == "key" else orc.convert_value_batched
        obmp, oidx, onzs = conv_o(orc.prune_magnitude(x_h, s))
        assert np.array_equal(arena.bmp[h, :t2].cpu().numpy(), obmp[0]) and np.array_equal(arena.idx[h, :t2 + 1].cpu().numpy(), oidx[0])
        assert np.array_equal(arena.nz[h, :int(used[h])].cpu().numpy().view(np.uint16), np.asarray(onzs[0]).view(np.uint16))
        del pruned, bmp, idx, nzs
    del K, V, ka, va
    torch.cuda.empty_cache()
