"""GPU: the device-resident cache container (mustafar_amd/cache.py) -- in-place append of the reference's cache-append
logic (models/llama_mustafar_kernel.py:339-390).  Bars: bitmaps, offsets and streams after appends are BIT-EXACT equal to
the reference-format compression of the concatenated tokens (oracle) and to the hook's tensor-op append; decode through a
cache view gives the same bits as decode through the contiguous layout."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pruned(heads, t, s, seed):
    from mustafar_amd import compression
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn((heads, t, 128), device=DEV, generator=g).half()
    return compression.prune_magnitude(x, s)


def _assert_same_as_oracle(arena, x_all, which):
    conv = orc.convert_key_batched if which == "key" else orc.convert_value_batched
    bmp, idx, nzs = conv(x_all.cpu().numpy())
    got = arena.to_reference()
    assert np.array_equal(got[0].cpu().numpy(), bmp)
    assert np.array_equal(got[1].cpu().numpy(), idx)
    assert len(got[2]) == len(nzs)
    for a, b in zip(got[2], nzs):
        assert np.array_equal(a.cpu().numpy().view(np.uint16), np.asarray(b).view(np.uint16))
    assert np.array_equal(got[3].cpu().numpy(), orc.nz_offset_from_idx(idx))


@pytest.mark.parametrize("which", ["key", "value"])
@pytest.mark.parametrize("heads,s", [(3, 0.7), (8, 0.5)])
def test_appends_match_compression_of_the_concatenation(which, heads, s):
    from mustafar_amd.cache import CompressedArena
    x1, x2, x3 = _pruned(heads, 512, s, 1), _pruned(heads, 256, s, 2), _pruned(heads, 64, s, 3)
    arena = CompressedArena.from_pruned(x1, which)
    _assert_same_as_oracle(arena, x1, which)
    arena.append(x2)
    arena.append(x3)
    assert arena.tokens == 832
    _assert_same_as_oracle(arena, torch.cat([x1, x2, x3], 1), which)


@pytest.mark.parametrize("which", ["key", "value"])
def test_growth_of_rows_and_stream_regions(which):
    """Capacities far too small on purpose: token rows and stream regions are re-housed, contents stay exact."""
    from mustafar_amd.cache import CompressedArena
    heads = 4
    arena = CompressedArena(heads, which, torch.device(DEV), cap_tokens=256, nz_cap=4096)
    parts = [_pruned(heads, 256, 0.5, 10 + i) for i in range(4)]
    for i, x in enumerate(parts):
        arena.append(x)
        _assert_same_as_oracle(arena, torch.cat(parts[: i + 1], 1), which)
    assert arena.cap_tokens >= 1024 and arena.nz_cap >= int(arena.used.max())
    # dense input (nothing pruned, every tile full) and an all-zero block in the same cache
    dense = torch.randn((heads, 64, 128), device=DEV).half()
    dense[dense == 0] = 1
    arena.append(dense)
    arena.append(torch.zeros((heads, 64, 128), device=DEV, dtype=torch.float16))
    _assert_same_as_oracle(arena, torch.cat(parts + [dense, torch.zeros_like(dense)], 1), which)


def test_from_reference_round_trip_and_hook_append_equivalence():
    """Arena append == the hook's tensor-op append (append_compressed, model :339-390) on the reference layout."""
    from mustafar_amd.cache import CompressedArena
    from mustafar_amd.hook import _compress, append_compressed
    heads = 6
    x1, x2 = _pruned(heads, 512, 0.7, 5), _pruned(heads, 256, 0.7, 6)
    ref = append_compressed(_compress(x1, "value"), _compress(x2, "value"), heads, 512, 256, 128)
    arena = CompressedArena.from_reference(_compress(x1, "value"), "value", 512)
    arena.append(x2)
    got = arena.to_reference()
    assert torch.equal(got[0].flatten(), ref[0].flatten()) and torch.equal(got[1].flatten(), ref[1].flatten())
    assert torch.equal(got[3], ref[3])
    for a, b in zip(got[2], ref[2]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("hq,hkv", [(8, 2), (4, 4), (8, 4)])
def test_decode_through_arena_is_bit_identical(hq, hkv):
    """Same prompt, same steps, across a 256-token trigger: arena cache vs contiguous cache, fused entry point."""
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    from mustafar_amd.cache import CompressedArena
    torch.manual_seed(3)
    bsz, D, L0 = 2, 128, 300
    K = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    V = torch.randn(bsz, hkv, L0, D, device=DEV).half()
    attns = [MustafarAttention(MustafarConfig(num_attention_heads=hq, num_key_value_heads=hkv, api="fused", arena=a))
             for a in (False, True)]
    pasts = [a.to_fused(a.build_cache(K.clone(), V.clone())) for a in attns]
    assert isinstance(pasts[1][0], CompressedArena) and not isinstance(pasts[0][0], CompressedArena)
    for step in range(262):
        qn = torch.randn(bsz, hq, 1, D, device=DEV).half()
        kn = torch.randn(bsz, hkv, 1, D, device=DEV).half()
        vn = torch.randn(bsz, hkv, 1, D, device=DEV).half()
        outs = []
        for i in (0, 1):
            o, pasts[i] = attns[i].decode(qn, kn, vn, pasts[i])
            outs.append(o)
        assert torch.equal(outs[0], outs[1]), f"step {step}"
    assert pasts[0][4] == pasts[1][4] == 512 and pasts[1][0].total_tokens == 512
    ref = pasts[1][0].to_reference()
    assert torch.equal(ref[0].flatten(), pasts[0][0][0].flatten()) and torch.equal(ref[1].flatten(), pasts[0][0][1].flatten())


def test_view_argument_checks():
    from mustafar_amd import _lib
    from mustafar_amd.cache import CompressedArena
    L = _lib.load()
    arena = CompressedArena.from_pruned(_pruned(2, 128, 0.7, 9), "key", cap_tokens=256)
    x = _pruned(2, 256, 0.7, 11)
    st = torch.cuda.current_stream().cuda_stream
    tot = torch.empty(2, dtype=torch.int64, device=DEV)
    # 128 + 256 tokens do not fit rows of 256 tokens: rejected on the host, nothing launched
    assert L.mustafar_cache_append_bitmap_key(st, x.data_ptr(), 2, 256, 128, arena.view_ptr(), 128, tot.data_ptr()) == 1
    assert L.mustafar_cache_append_bitmap_key(st, x.data_ptr(), 2, 256, 128, arena.view_ptr(), 100, tot.data_ptr()) == 1   # % 64
    assert L.mustafar_cache_append_pack_key(st, x.data_ptr(), 2, 256, 128, None, 128) == 1


# ---- fused forms: prune + compress + append of RAW rows of K and V together (mustafar_cache_append_kv) ----------------
def _raw(B, H, rows, seed, special=False):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn((B, H, rows, 128), device=DEV, generator=g).half()
    if special:   # ties at the threshold, signed zeros, an all-zero row, a constant row (everything ties: all 128 kept)
        x[0, 0, 0, :] = 0
        x[0, 0, 1, :] = 0.5
        x[0, 0, 2, ::2] = x[0, 0, 2, 1::2]
        x[0, 0, 3, :8] = -0.0
        x[-1, -1, 5, :] = torch.tensor([1.0, -1.0] * 64, device=DEV).half()
    return x


def _oracle_pruned(x, s, t):
    B, H, _, D = x.shape
    return torch.from_numpy(orc.prune_magnitude(x[:, :, :t].contiguous().cpu().numpy(), s)).reshape(B * H, t, D)


@pytest.mark.parametrize("s_k,s_v", [(0.7, 0.7), (0.5, 0.8)])
def test_fused_prefill_from_raw_rows_matches_oracle_prune_and_compress(s_k, s_v):
    """from_raw_pair reads RAW K/V [B, Hkv, L, 128] (head stride L * 128, only the first t tokens of every head) and must
    equal oracle-prune + oracle-compress bit for bit -- thresholds, ties, signed zeros included."""
    from mustafar_amd import compression
    from mustafar_amd.cache import CompressedArena
    B, H, L, t = 2, 3, 600, 512
    K, V = _raw(B, H, L, 21, special=True), _raw(B, H, L, 22, special=True)
    ka, va = CompressedArena.from_raw_pair(K, V, t, compression.kth_from_sparsity(s_k, 128), compression.kth_from_sparsity(s_v, 128))
    assert ka.tokens == va.tokens == t
    _assert_same_as_oracle(ka, _oracle_pruned(K, s_k, t), "key")
    _assert_same_as_oracle(va, _oracle_pruned(V, s_v, t), "value")


def test_fused_prefill_survives_an_underestimated_stream_region():
    """Constant rows tie at the threshold, so every element is kept (model :107): three times the estimate.  The first attempt
    raises bit 0 of the device flag and reports the length every head needs; the launch is repeated at that size."""
    from mustafar_amd.cache import CompressedArena
    K = torch.full((1, 2, 256, 128), 0.25, device=DEV, dtype=torch.float16)
    V = -K
    ka, va = CompressedArena.from_raw_pair(K, V, 256, 89, 89)
    assert int(ka.used.min()) == 256 * 128 and int(va.used.min()) == 256 * 128
    _assert_same_as_oracle(ka, K.reshape(2, 256, 128), "key")
    _assert_same_as_oracle(va, V.reshape(2, 256, 128), "value")


def test_fused_trigger_append_from_a_window_buffer():
    """The decode trigger (model :324-398): rows [0, 256) of window buffers with spare rows, appended behind 512 tokens,
    twice (the second append sizes its room from the first one's asynchronously delivered stream lengths), then the slide."""
    from mustafar_amd import compression
    from mustafar_amd.cache import CompressedArena
    from mustafar_amd.hook import Window
    B, H, s = 2, 2, 0.7
    kth = compression.kth_from_sparsity(s, 128)
    K0, V0 = _raw(B, H, 512, 31), _raw(B, H, 512, 32)
    ka, va = CompressedArena.from_raw_pair(K0, V0, 512, kth, kth, cap_tokens=768)
    allK, allV = [_oracle_pruned(K0, s, 512)], [_oracle_pruned(V0, s, 512)]
    for i in range(3):   # the third append outgrows the 768-token rows: re-housed, contents stay exact
        kw, vw = Window(_raw(B, H, 288, 40 + i, special=(i == 1)), 352), Window(_raw(B, H, 288, 50 + i), 352)
        keep_k, keep_v = kw.buf[:, :, 256:288].clone(), vw.buf[:, :, 256:288].clone()
        allK.append(_oracle_pruned(kw.buf, s, 256))
        allV.append(_oracle_pruned(vw.buf, s, 256))
        CompressedArena.append_window_pair(ka, va, kw.buf, vw.buf, 256, kth, kth)
        Window.drop_front_pair(kw, vw, 256)
        assert kw.len == vw.len == 32
        assert torch.equal(kw.view(), keep_k) and torch.equal(vw.view(), keep_v)
    assert ka.tokens == va.tokens == 1280
    _assert_same_as_oracle(ka, torch.cat(allK, 1), "key")
    _assert_same_as_oracle(va, torch.cat(allV, 1), "value")


def test_fused_append_equals_the_two_call_form():
    """Same tokens through prune_magnitude + CompressedArena.append (the round-1 path, one host read per side) and through
    append_window_pair: identical bytes in the arena."""
    from mustafar_amd import compression
    from mustafar_amd.cache import CompressedArena
    B, H, s = 1, 4, 0.8
    kth = compression.kth_from_sparsity(s, 128)
    X0, X1 = _raw(B, H, 256, 61), _raw(B, H, 320, 62)
    a_k, a_v = CompressedArena.from_raw_pair(X0, X0, 256, kth, kth, cap_tokens=1024)
    CompressedArena.append_window_pair(a_k, a_v, X1, X1, 256, kth, kth)
    for which, fused in (("key", a_k), ("value", a_v)):
        b = CompressedArena.from_pruned(compression.prune_magnitude(X0.reshape(H, 256, 128), s), which, cap_tokens=1024)
        b.append(compression.prune_magnitude(X1[:, :, :256].reshape(H, 256, 128).contiguous(), s))
        fr, br = fused.to_reference(), b.to_reference()
        assert torch.equal(fr[0], br[0]) and torch.equal(fr[1], br[1]) and torch.equal(fr[3], br[3])
        assert torch.equal(torch.cat(fr[2]).view(torch.int16), torch.cat(br[2]).view(torch.int16))


def test_fused_append_replays_from_a_captured_graph():
    """mustafar_cache_append_kv (memset node + one launch) and the window slide captured in a hipGraph: nothing is allocated
    and nothing is read back inside the call, so the replay appends whatever the window buffers hold at that moment.  The
    graph writes behind the same 256 tokens every time it runs: replayed on two different windows, each result must equal
    the eager call on the same data, bit for bit."""
    import ctypes
    from mustafar_amd import _lib, compression
    from mustafar_amd.cache import CompressedArena
    B, H, s = 2, 2, 0.7
    kth = compression.kth_from_sparsity(s, 128)
    L = _lib.load()
    K0, V0 = _raw(B, H, 256, 71), _raw(B, H, 256, 72)

    def fresh():
        return CompressedArena.from_raw_pair(K0, V0, 256, kth, kth, cap_tokens=1024)

    ka, va = fresh()
    for a in (ka, va):   # room for one worst-case append behind the 256 tokens in use (what append_window_pair secures on the host)
        if int(a.used.max()) + 256 * 128 > a.nz_cap:
            a._rehouse(a.cap_tokens, int(a.used.max()) + 256 * 128 + 1024)
    kw = torch.zeros((B, H, 352, 128), dtype=torch.float16, device=DEV)
    vw = torch.zeros_like(kw)
    scratch = torch.empty(int(L.mustafar_compress_scratch_bytes(B * H, 256)), dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    k_tot = torch.zeros(B * H, dtype=torch.int64, device=DEV)
    v_tot = torch.zeros_like(k_tot)

    def call(stream):
        _lib.check(L.mustafar_cache_append_kv(stream, kw.data_ptr(), vw.data_ptr(), 352 * 128, B * H, 256, 128, kth, kth, ka.view_ptr(), va.view_ptr(),
                                              256, k_tot.data_ptr(), v_tot.data_ptr(), ka.nz_cap, va.nz_cap, flag.data_ptr(), scratch.data_ptr()), "append_kv")
        _lib.check(L.mustafar_window_drop_front(stream, kw.data_ptr(), vw.data_ptr(), 352 * 128, B * H, 288, 256), "drop_front")

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call(torch.cuda.current_stream().cuda_stream)
    for seed in (81, 82):
        wk, wv = _raw(B, H, 288, seed, special=(seed == 82)), _raw(B, H, 288, seed + 10)
        kw[:, :, :288] = wk
        vw[:, :, :288] = wv
        g.replay()
        torch.cuda.synchronize()
        assert int(flag) == 0
        assert torch.equal(kw[:, :, :32], wk[:, :, 256:288]) and torch.equal(vw[:, :, :32], wv[:, :, 256:288])
        ea, eb = fresh()
        CompressedArena.append_window_pair(ea, eb, wk.contiguous(), wv.contiguous(), 256, kth, kth)
        for got, want, tot in ((ka, ea, k_tot), (va, eb, v_tot)):
            t = 512 * 2
            assert torch.equal(got.bmp[:, :t], want.bmp[:, :t]) and torch.equal(got.idx[:, :t + 1], want.idx[:, :t + 1])
            assert torch.equal(tot.cpu(), want.used.cpu() if torch.is_tensor(want.used) else torch.as_tensor(want.used))
            for h in range(B * H):
                n = int(tot[h])
                assert torch.equal(got.nz[h, :n].view(torch.int16), want.nz[h, :n].view(torch.int16))


def test_one_pass_and_two_pass_compression_write_the_same_bytes():
    """MUSTAFAR_COMPRESS=twopass (pass 1 + scan + pass 2, read once per process) in a child process against the default
    one-pass kernel here: same raw rows -> identical bitmaps, offsets and streams."""
    import os
    import subprocess
    import sys
    import tempfile
    from mustafar_amd import compression
    from mustafar_amd.cache import CompressedArena
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from tests.test_gpu_cache import _raw
from mustafar_amd import compression
from mustafar_amd.cache import CompressedArena
K, V = _raw(2, 3, 600, 91, special=True), _raw(2, 3, 600, 92)
ka, va = CompressedArena.from_raw_pair(K, V, 512, 89, 102)
torch.save([ [x.cpu() for x in a.to_reference()[:2]] + [torch.cat(a.to_reference()[2]).cpu()] for a in (ka, va)], sys.argv[1])
''' % root
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "twopass.pt")
        env = dict(os.environ, MUSTAFAR_COMPRESS="twopass")
        subprocess.check_call([sys.executable, "-c", code, out], env=env, cwd=root)
        want = torch.load(out)
    K, V = _raw(2, 3, 600, 91, special=True), _raw(2, 3, 600, 92)
    ka, va = CompressedArena.from_raw_pair(K, V, 512, 89, 102)
    for a, w in zip((ka, va), want):
        r = a.to_reference()
        assert torch.equal(r[0].cpu(), w[0]) and torch.equal(r[1].cpu(), w[1])
        assert torch.equal(torch.cat(r[2]).cpu().view(torch.int16), w[2].view(torch.int16))


@pytest.mark.parametrize("B,H,t", [(1, 1, 64), (1, 3, 192), (2, 1, 1024)])
def test_one_pass_compression_of_pruned_rows_and_small_shapes(B, H, t):
    """kth = 0 (rows already pruned: no threshold search) through the one-pass kernel, at one block per head, an odd number of
    heads and sixteen blocks per head (the blocks of a head find their stream positions from each other): equal to the two-call
    conversion of the same rows."""
    from mustafar_amd import compression
    from mustafar_amd.cache import CompressedArena
    X = compression.prune_magnitude(_raw(B, H, t, 97, special=(t >= 192)).reshape(B * H, t, 128), 0.7).reshape(B, H, t, 128)
    ka, va = CompressedArena.from_raw_pair(X, X, t, 0, 0)
    for which, arena in (("key", ka), ("value", va)):
        want = CompressedArena.from_pruned(X.reshape(B * H, t, 128), which)
        a, w = arena.to_reference(), want.to_reference()
        assert torch.equal(a[0], w[0]) and torch.equal(a[1], w[1]) and torch.equal(a[3], w[3])
        assert torch.equal(torch.cat(a[2]).view(torch.int16), torch.cat(w[2]).view(torch.int16))


@pytest.mark.parametrize("kth", [0, 1])
def test_one_pass_compression_of_rows_without_a_single_zero(kth):
    """Nothing to prune and no zero anywhere: every tile holds 64 values, so each half of a block's stream fills the kernel's 8 KB
    image to the last byte (the image is sized for exactly this).  Equal to the two-call conversion of the same rows."""
    from mustafar_amd.cache import CompressedArena
    g = torch.Generator(device="cpu").manual_seed(5)
    X = (torch.rand((1, 3, 192, 128), generator=g) + 0.5) * (torch.randint(0, 2, (1, 3, 192, 128), generator=g) * 2 - 1)
    X = X.half().cuda()
    assert int((X == 0).sum()) == 0
    ka, va = CompressedArena.from_raw_pair(X, X, 192, kth, kth)
    for which, arena in (("key", ka), ("value", va)):
        want = CompressedArena.from_pruned(X.reshape(3, 192, 128), which)
        a, w = arena.to_reference(), want.to_reference()
        assert torch.equal(a[0], w[0]) and torch.equal(a[1], w[1]) and torch.equal(a[3], w[3])
        assert torch.equal(torch.cat(a[2]).view(torch.int16), torch.cat(w[2]).view(torch.int16))
        assert int(torch.cat(a[2]).numel()) == 3 * 192 * 128   # nothing dropped, nothing padded


@pytest.mark.parametrize("seed", range(6))
def test_one_pass_compression_randomised_against_the_oracle(seed):
    """Random shapes, sparsities and value distributions (quantised values: many ties at the threshold; blocks of zeros; a few
    huge and a few subnormal entries) through prune + compress in one pass, K and V with different k, against oracle prune +
    oracle compress bit for bit."""
    from mustafar_amd import compression
    from mustafar_amd.cache import CompressedArena
    rng = np.random.default_rng(1000 + seed)
    B, H = int(rng.integers(1, 3)), int(rng.integers(1, 5))
    t = 64 * int(rng.integers(1, 9))
    L = t + int(rng.integers(0, 3)) * 32
    s_k, s_v = float(rng.choice([0.3, 0.5, 0.7, 0.8, 0.9])), float(rng.choice([0.3, 0.5, 0.7, 0.8, 0.9]))

    def make():
        x = rng.standard_normal((B, H, L, 128)).astype(np.float32)
        mode = rng.integers(0, 4)
        if mode == 1:
            x = np.round(x * 2) / 2                      # heavy ties
        elif mode == 2:
            x[rng.random(x.shape) < 0.6] = 0.0           # already sparse: thresholds of zero, empty tiles
        elif mode == 3:
            x *= 10.0 ** rng.integers(-7, 4, size=(B, H, L, 1))   # rows from subnormal to large
        x[rng.random(x.shape) < 0.01] *= -0.0
        return torch.from_numpy(np.clip(x, -65504, 65504).astype(np.float16)).to(DEV)

    K, V = make(), make()
    ka, va = CompressedArena.from_raw_pair(K, V, t, compression.kth_from_sparsity(s_k, 128), compression.kth_from_sparsity(s_v, 128))
    _assert_same_as_oracle(ka, _oracle_pruned(K, s_k, t), "key")
    _assert_same_as_oracle(va, _oracle_pruned(V, s_v, t), "value")


# ---- round 3: sizing, failure paths ---------------------------------------------------------------------------------------
def test_arena_is_housed_within_its_slack_and_appends_keep_it_there():
    """Reserved bytes stay within (1 + slack) x 1.02 of the bytes in use (+ the rounding of the rows to 64 tokens and 1 KB per head)
    from the prefill on and across triggers -- the half of the metric that counts allocated bytes -- and the contents stay exact
    through every re-housing (at this size every trigger re-houses)."""
    from mustafar_amd import compression
    from mustafar_amd.cache import CompressedArena, DEFAULT_SLACK
    from mustafar_amd.hook import Window
    B, H, s = 1, 4, 0.7
    kth = compression.kth_from_sparsity(s, 128)
    K0, V0 = _raw(B, H, 768, 101), _raw(B, H, 768, 102)
    ka, va = CompressedArena.from_raw_pair(K0, V0, 768, kth, kth)
    allK, allV = [_oracle_pruned(K0, s, 768)], [_oracle_pruned(V0, s, 768)]

    def tight(a):
        rows = 24 * a.heads * 64          # rows are rounded up to 64 tokens (24 B of bitmaps + offsets per token)
        return a.bytes_reserved() <= a.bytes_in_use() * (1 + DEFAULT_SLACK) * 1.02 + rows + 1024 * a.heads

    assert tight(ka) and tight(va)
    for i in range(5):
        kw, vw = Window(_raw(B, H, 288, 110 + i), 288), Window(_raw(B, H, 288, 120 + i), 288)
        allK.append(_oracle_pruned(kw.buf, s, 256))
        allV.append(_oracle_pruned(vw.buf, s, 256))
        CompressedArena.append_window_pair(ka, va, kw.buf, vw.buf, 256, kth, kth)
        assert ka.tokens == va.tokens == 768 + 256 * (i + 1)
        assert tight(ka) and tight(va), (i, ka.bytes_reserved(), ka.bytes_in_use())
    _assert_same_as_oracle(ka, torch.cat(allK, 1), "key")
    _assert_same_as_oracle(va, torch.cat(allV, 1), "value")


def test_trigger_append_that_outgrows_the_expected_room_is_repeated_at_the_measured_size():
    """An append of rows full of ties (every element kept: 128 halfs per token against the ~48 the cache's history predicts)
    sets bit 0 of the device flag; the host re-houses at the lengths the launch reported and repeats it.  Exact afterwards."""
    from mustafar_amd import compression
    from mustafar_amd.cache import CompressedArena
    B, H, s = 1, 3, 0.7
    kth = compression.kth_from_sparsity(s, 128)
    K0, V0 = _raw(B, H, 512, 131), _raw(B, H, 512, 132)
    ka, va = CompressedArena.from_raw_pair(K0, V0, 512, kth, kth)
    before = (ka.nz_cap, va.nz_cap)
    kw = torch.full((B, H, 288, 128), 0.75, device=DEV, dtype=torch.float16)
    vw = _raw(B, H, 288, 133)
    CompressedArena.append_window_pair(ka, va, kw, vw, 256, kth, kth)
    assert ka.tokens == va.tokens == 768 and ka.nz_cap > before[0] and int(ka._overflow) == 0
    assert int(ka.used.min()) >= 256 * 128
    _assert_same_as_oracle(ka, torch.cat([_oracle_pruned(K0, s, 512), kw[:, :, :256].reshape(H, 256, 128).cpu()], 1), "key")
    _assert_same_as_oracle(va, torch.cat([_oracle_pruned(V0, s, 512), _oracle_pruned(vw, s, 256)], 1), "value")


def test_a_compression_timeout_is_answered_by_one_repeat_in_the_two_pass_form():
    """Bit 1 of the device flag (a block gave up waiting for its predecessors' lengths): the flag is pre-set here, as if a block of this very
    launch had set it (tests/test_gpu_extents.py provokes the real thing through mustafar_compress_test_skip_publish).  Round 4 raised
    ArenaAppendTimeout and left the cache as it was; round 5 repeats the append once in the two-pass form -- the raw rows are still in
    place, the call is idempotent -- counts it, and the cache holds exactly what an undisturbed append gives."""
    from mustafar_amd import cache, compression
    from mustafar_amd.cache import CompressedArena
    B, H, s = 1, 2, 0.7
    kth = compression.kth_from_sparsity(s, 128)
    K0, V0 = _raw(B, H, 256, 141), _raw(B, H, 256, 142)
    ka, va = CompressedArena.from_raw_pair(K0, V0, 256, kth, kth)
    kw, vw = _raw(B, H, 288, 143), _raw(B, H, 288, 144)
    for a in (ka, va):                                    # room as the call itself would make it, so that the flag tensor stays the one pre-set
        a._make_room(256, int(a.used.max()) + a._expected_append(256, kth))
    va._overflow = ka._overflow
    ka._overflow.fill_(2)
    before = cache.compress_fallbacks
    CompressedArena.append_window_pair(ka, va, kw, vw, 256, kth, kth)
    assert cache.compress_fallbacks == before + 1
    assert ka.tokens == va.tokens == 512 and int(ka._overflow) == 0
    _assert_same_as_oracle(ka, torch.cat([_oracle_pruned(K0, s, 256), _oracle_pruned(kw, s, 256)], 1), "key")
    _assert_same_as_oracle(va, torch.cat([_oracle_pruned(V0, s, 256), _oracle_pruned(vw, s, 256)], 1), "value")


def test_the_one_pass_form_refuses_to_run_without_a_flag():
    from mustafar_amd import _lib, compression
    from mustafar_amd.cache import CompressedArena
    L = _lib.load()
    K0 = _raw(1, 2, 256, 151)
    ka, va = CompressedArena.from_raw_pair(K0, K0, 256, 89, 89, cap_tokens=1024)
    kw = _raw(1, 2, 288, 152)
    scratch = torch.empty(int(L.mustafar_compress_scratch_bytes(2, 256)), dtype=torch.uint8, device=DEV)
    tot = torch.zeros(2, dtype=torch.int64, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert L.mustafar_cache_append_kv(st, kw.data_ptr(), kw.data_ptr(), 288 * 128, 2, 256, 128, 89, 89, ka.view_ptr(), va.view_ptr(), 256,
                                      tot.data_ptr(), tot.data_ptr(), ka.nz_cap, va.nz_cap, None, scratch.data_ptr()) == 1


def test_prefill_from_transposed_layout_k_and_v():
    """In the model K and V reach the hook as transpose(1, 2) views of [B, L, H, D] projections (RoPE keeps the strides): the
    arena prefill takes them too (through a contiguous copy) and builds the same cache as from contiguous tensors."""
    from mustafar_amd.cache import CompressedArena
    from mustafar_amd.hook import MustafarAttention, MustafarConfig
    torch.manual_seed(161)
    B, Hkv, L, D = 2, 2, 600, 128
    Kt = torch.randn(B, L, Hkv, D, device=DEV).half().transpose(1, 2)
    Vt = torch.randn(B, L, Hkv, D, device=DEV).half().transpose(1, 2)
    assert not Kt.is_contiguous()
    attn = MustafarAttention(MustafarConfig(num_attention_heads=8, num_key_value_heads=Hkv, api="fused", arena=True))
    past = attn.build_cache(Kt, Vt)
    assert isinstance(past[0], CompressedArena) and past[4] == 512
    _assert_same_as_oracle(past[0], _oracle_pruned(Kt, 0.7, 512), "key")
    _assert_same_as_oracle(past[2], _oracle_pruned(Vt, 0.7, 512), "value")
    assert torch.equal(past[1], Kt[:, :, 512:]) and torch.equal(past[3], Vt[:, :, 512:])
