"""GPU: HIP path (through the C ABI / mustafar_package mirror) against the CPU oracle and golden fixtures.

Integer/byte results (prune, bitmaps, offsets, packed stream) must be bit-exact.  The two SpMV results are
fp16 roundings of fp32 sums: they must sit within `fp16_bound` of the oracle's float64 sums (tolerance
stated in tests/util.py) -- the same bound the oracle's own fp16 output satisfies.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.util import fp16_bound, make_cache

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    from mustafar_amd import compression, mustafar_package
    return mustafar_package, compression


@pytest.fixture(params=["valu", "mfma", "valu-other-forms"], autouse=True)
def fma_engine(request):
    """Every test of this module runs on both FMA engines of the GQA-4 kernels, and once more with the kernel forms
    that are NOT the default of the two entry points (round 4: the key entry point defaults to the lean pair kernel, the value entry
    point to round 1's kernel; mustafar_tune(6 / 7) select the other one of each)."""
    from mustafar_amd import _lib
    L = _lib.load()
    assert L.mustafar_set_fma_engine(1 if request.param == "mfma" else 0) == 0
    other = request.param.endswith("other-forms")
    assert L.mustafar_tune(6, 0 if other else 1) == 0 and L.mustafar_tune(7, 1 if other else 0) == 0   # (value: forced lean / forced round-1 kernel)
    yield request.param
    L.mustafar_set_fma_engine(2)   # the process defaults
    L.mustafar_tune(6, 1)
    L.mustafar_tune(7, 2)   # (by size: the default since round 5)


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(DEV) if dtype is None else t.to(DEV, dtype)


def _cache_to_dev(c):
    return (_t(c["bmp"]), _t(np.concatenate(c["nzs"]) if len(c["nzs"]) else np.zeros(0, np.float16)),
            _t(c["idx"]), _t(c["nz_offset"]))


def _check_spmv(got, C16, Cd, sumabs, what):
    got = got.float().cpu().numpy().astype(np.float64)
    bound = fp16_bound(Cd, sumabs)
    err = np.abs(got - Cd)
    assert (err <= bound).all(), f"{what}: max excess {np.max(err - bound):.3e} at {np.unravel_index(np.argmax(err - bound), err.shape)}"
    assert (np.abs(C16.astype(np.float64) - Cd) <= bound).all(), "oracle itself violates the bound"


@pytest.mark.parametrize("groups,N,B,t,s,adv", [
    (1, 8, 2, 256, 0.7, False), (4, 8, 2, 256, 0.7, False), (4, 1, 3, 320, 0.5, False), (2, 1, 2, 64, 0.8, False),
    (1, 1, 1, 1024, 0.7, False), (4, 8, 1, 512, 0.8, True), (8, 1, 1, 192, 0.7, False), (3, 8, 2, 128, 0.7, False),
])
def test_key_spmv_vs_oracle(pkg, groups, N, B, t, s, adv):
    mp, _ = pkg
    c = make_cache("key", B, t, 128, s, seed=100 + t + groups, adversarial=adv)
    BH = B * groups
    rng = np.random.default_rng(5)
    q = np.zeros((BH, N, 128), np.float16)
    q[:, 0] = rng.standard_normal((BH, 128)).astype(np.float16)          # hook layout: row 0 real, rows 1.. zero
    bmp, nz, idx, off = _cache_to_dev(c)
    out = mp.mustafar_key_formulation(bmp, nz, idx, off, _t(q), t, 128, BH, groups)
    assert out.shape == (BH, N, t) and out.dtype == torch.float16
    C16, Cd = orc.key_spmv(c["bmp"], np.concatenate(c["nzs"]), c["idx"], c["nz_offset"], q, t, 128, BH, groups)
    K = c["pruned"].astype(np.float64)
    sumabs = np.stack([np.abs(K[b // groups]) @ np.abs(q[b].astype(np.float64)).T for b in range(BH)]).transpose(0, 2, 1)
    _check_spmv(out, C16, Cd, sumabs, "key")
    if N == 8:
        assert not out[:, 1:].any(), "pad rows must be exact zeros"


@pytest.mark.parametrize("groups", [1, 4])
def test_key_spmv_nonzero_pad_rows(pkg, groups):
    """The reference computes all 8 rows of the padded query; so must we when they are not zero."""
    mp, _ = pkg
    B, t = 2, 256
    c = make_cache("key", B, t, 128, 0.7, seed=77)
    BH = B * groups
    rng = np.random.default_rng(6)
    q = rng.standard_normal((BH, 8, 128)).astype(np.float16)
    q[:, 2] = 0
    q[0, 5] = 0
    q[:, 6, ::2] = np.float16(-0.0)
    bmp, nz, idx, off = _cache_to_dev(c)
    out = mp.mustafar_key_formulation(bmp, nz, idx, off, _t(q), t, 128, BH, groups)
    C16, Cd = orc.key_spmv(c["bmp"], np.concatenate(c["nzs"]), c["idx"], c["nz_offset"], q, t, 128, BH, groups)
    K = c["pruned"].astype(np.float64)
    sumabs = np.stack([np.abs(K[b // groups]) @ np.abs(q[b].astype(np.float64)).T for b in range(BH)]).transpose(0, 2, 1)
    _check_spmv(out, C16, Cd, sumabs, "key-8rows")


@pytest.mark.parametrize("groups,N,B,t,s,split,adv", [
    (1, 8, 2, 256, 0.7, 0, False), (4, 8, 2, 256, 0.7, 0, False), (4, 1, 3, 320, 0.5, 0, False),
    (2, 1, 2, 64, 0.8, 0, False), (1, 1, 1, 1024, 0.7, 1, False), (4, 1, 1, 1024, 0.7, 3, False),
    (4, 8, 1, 512, 0.8, 0, True), (8, 1, 1, 192, 0.7, 2, False), (3, 8, 2, 128, 0.7, 1, False),
    (4, 8, 1, 1024, 0.7, 5, False),
])
def test_value_spmv_vs_oracle(pkg, groups, N, B, t, s, split, adv):
    mp, _ = pkg
    c = make_cache("value", B, t, 128, s, seed=200 + t + groups, adversarial=adv)
    BH = B * groups
    rng = np.random.default_rng(8)
    p = np.zeros((BH, N, t), np.float16)
    logits = rng.standard_normal((BH, t)) * 2
    p[:, 0] = (np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)).astype(np.float16)
    bmp, nz, idx, off = _cache_to_dev(c)
    ws = torch.zeros(1, dtype=torch.float16, device=DEV)             # the model's 1-element workspace (:658)
    out = mp.mustafar_value_formulation(bmp, nz, idx, off, _t(p), ws, 128, t, BH, groups, split_k=split)
    assert out.shape == (BH, N, 128) and out.dtype == torch.float16
    C16, Cd = orc.value_spmv(c["bmp"], np.concatenate(c["nzs"]), c["idx"], c["nz_offset"], p, 128, t, BH, groups)
    V = c["pruned"].astype(np.float64)
    sumabs = np.stack([np.abs(p[b].astype(np.float64)) @ np.abs(V[b // groups]) for b in range(BH)])
    _check_spmv(out, C16, Cd, sumabs, "value")
    if N == 8:
        assert not out[:, 1:].any()


@pytest.mark.parametrize("groups,split", [(1, 0), (4, 0), (4, 1), (2, 4)])
def test_value_spmv_nonzero_pad_rows(pkg, groups, split):
    mp, _ = pkg
    B, t = 2, 512
    c = make_cache("value", B, t, 128, 0.7, seed=99)
    BH = B * groups
    rng = np.random.default_rng(9)
    p = (rng.random((BH, 8, t)) / t).astype(np.float16)
    p[:, 3] = 0
    p[:, 4, : t // 2] = 0            # non-zero only in the second half of the tokens: per-chunk row masks differ
    p[1, 7] = 0
    bmp, nz, idx, off = _cache_to_dev(c)
    ws = torch.zeros(1, dtype=torch.float16, device=DEV)
    out = mp.mustafar_value_formulation(bmp, nz, idx, off, _t(p), ws, 128, t, BH, groups, split_k=split)
    C16, Cd = orc.value_spmv(c["bmp"], np.concatenate(c["nzs"]), c["idx"], c["nz_offset"], p, 128, t, BH, groups)
    V = c["pruned"].astype(np.float64)
    sumabs = np.stack([np.abs(p[b].astype(np.float64)) @ np.abs(V[b // groups]) for b in range(BH)])
    _check_spmv(out, C16, Cd, sumabs, "value-8rows")


@pytest.mark.parametrize("groups,t,split", [(4, 1408, 11), (4, 1408, 6), (2, 704, 11), (1, 1408, 22)])
def test_value_spmv_pad_rows_live_in_single_chunks(pkg, groups, t, split):
    """Round 6: the lean form reads the pad rows in workgroups of their own, one per head group and FOUR token chunks, a byte of row mask per
    chunk.  Pad rows that hold a non-zero in exactly one chunk -- the first, one in the middle of a group, the last column of the last (partial)
    group -- and one head group with nothing: every slab's mask is its own, and the rows come out as the reference computes them."""
    mp, _ = pkg
    B = 2
    c = make_cache("value", B, t, 128, 0.7, seed=123)
    BH = B * groups
    rng = np.random.default_rng(10)
    p = np.zeros((BH, 8, t), np.float16)
    p[:, 0] = (rng.random((BH, t)) / t).astype(np.float16)
    p[0, 1, 5] = 0.25                       # first chunk only
    p[0, 2, 130:140] = 0.125                # a chunk in the middle of the first group
    p[BH - 1, 5, t - 1] = 0.5               # the last column: last chunk of the last group
    p[groups - 1, 6] = (rng.random(t) / t).astype(np.float16)   # a whole row, last head of the first head group
    p[0, 7, 64 * 9] = -0.0                  # minus zero is zero
    bmp, nz, idx, off = _cache_to_dev(c)
    ws = torch.zeros(1, dtype=torch.float16, device=DEV)
    out = mp.mustafar_value_formulation(bmp, nz, idx, off, _t(p), ws, 128, t, BH, groups, split_k=split)
    C16, Cd = orc.value_spmv(c["bmp"], np.concatenate(c["nzs"]), c["idx"], c["nz_offset"], p, 128, t, BH, groups)
    V = c["pruned"].astype(np.float64)
    sumabs = np.stack([np.abs(p[b].astype(np.float64)) @ np.abs(V[b // groups]) for b in range(BH)])
    _check_spmv(out, C16, Cd, sumabs, "value-8rows-single-chunks")
    assert not out[:, 3:5].any() and not out[:, 7].any(), "rows without a non-zero are exact zeros"
    assert out[0, 1].any() and out[0, 2].any() and out[BH - 1, 5].any() and out[groups - 1, 6].any()


def test_value_spmv_eight_rows_replayed_from_a_graph_follows_the_pad_rows(pkg):
    """The 8-row value call captured ONCE (the workspace is the call's own allocation inside the capture) and replayed over operands that change in place: pad rows all zero,
    then a pad row live in one token chunk, then zero again.  Which rows a launch computes is decided on the device at every replay (the pad workgroups' masks), not
    at capture time."""
    mp, _ = pkg
    groups, B, t = 4, 2, 1024
    c = make_cache("value", B, t, 128, 0.7, seed=77)
    BH = B * groups
    rng = np.random.default_rng(12)
    p0 = np.zeros((BH, 8, t), np.float16)
    p0[:, 0] = (rng.random((BH, t)) / t).astype(np.float16)
    p1 = p0.copy()
    p1[3, 5, 700:720] = 0.25
    p1[:, 0] = (rng.random((BH, t)) / t).astype(np.float16)
    bmp, nz, idx, off = _cache_to_dev(c)
    ws = torch.zeros(1, dtype=torch.float16, device=DEV)
    p_dev = _t(p0)
    mp.mustafar_value_formulation(bmp, nz, idx, off, p_dev, ws, 128, t, BH, groups)   # (warm-up outside the capture)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = mp.mustafar_value_formulation(bmp, nz, idx, off, p_dev, ws, 128, t, BH, groups)
    V = c["pruned"].astype(np.float64)
    for p in (p0, p1, p0):
        p_dev.copy_(_t(p))
        g.replay()
        torch.cuda.synchronize()
        C16, Cd = orc.value_spmv(c["bmp"], np.concatenate(c["nzs"]), c["idx"], c["nz_offset"], p, 128, t, BH, groups)
        sumabs = np.stack([np.abs(p[b].astype(np.float64)) @ np.abs(V[b // groups]) for b in range(BH)])
        _check_spmv(out, C16, Cd, sumabs, "value-8rows-graph")
        live = np.abs(p[:, 1:]).sum(-1) > 0
        assert np.array_equal(out[:, 1:].abs().sum(-1).cpu().numpy() > 0, live), "exactly the pad rows that hold a non-zero come out non-zero"


def test_spmv_on_golden_compressed_streams(pkg, golden_dir):
    """Feed the REFERENCE-produced compressed tensors (Triton kernels, fixtures) straight to the HIP kernels."""
    mp, _ = pkg
    g = np.load(os.path.join(golden_dir, "compress_reference.npz"))
    rng = np.random.default_rng(11)
    for name in ("rand_B2_t256_s0.7", "edge_tiles", "rand_B1_t512_s0.7", "rand_B2_t128_s0.0"):
        x = g[f"{name}__x"].view(np.float16)
        B, t, D = x.shape
        for which in ("key", "value"):
            bmp, accum = g[f"{name}__{which}__bmp"], g[f"{name}__{which}__accum"]
            packed = g[f"{name}__{which}__packed"].view(np.float16)
            off = orc.nz_offset_from_idx(accum)
            groups = 2
            BH = B * groups
            if which == "key":
                q = rng.standard_normal((BH, 1, D)).astype(np.float16)
                out = mp.mustafar_key_formulation(_t(bmp), _t(packed), _t(accum), _t(off), _t(q), t, D, BH, groups)
                want = np.stack([x[b // groups].astype(np.float64) @ q[b, 0].astype(np.float64) for b in range(BH)])[:, None]
                sumabs = np.stack([np.abs(x[b // groups].astype(np.float64)) @ np.abs(q[b, 0].astype(np.float64)) for b in range(BH)])[:, None]
            else:
                p = (rng.random((BH, 1, t)) / t).astype(np.float16)
                ws = torch.zeros(1, dtype=torch.float16, device=DEV)
                out = mp.mustafar_value_formulation(_t(bmp), _t(packed), _t(accum), _t(off), _t(p), ws, D, t, BH, groups)
                want = np.stack([p[b, 0].astype(np.float64) @ x[b // groups].astype(np.float64) for b in range(BH)])[:, None]
                sumabs = np.stack([np.abs(p[b, 0].astype(np.float64)) @ np.abs(x[b // groups].astype(np.float64)) for b in range(BH)])[:, None]
            err = np.abs(out.float().cpu().numpy().astype(np.float64) - want)
            assert (err <= fp16_bound(want, sumabs)).all(), (name, which)


def test_prune_bit_exact(pkg, golden_dir):
    _, comp = pkg
    g = np.load(os.path.join(golden_dir, "prune_reference.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    checked = 0
    for name in names:
        x = g[f"{name}__x"]
        if x.shape[-1] != 128:
            continue
        s = float(g[f"{name}__s"])
        got = comp.prune_magnitude(_t(x.view(np.float16)), s).cpu().numpy().view(np.uint16)
        assert np.array_equal(got, g[f"{name}__y"]), name
        checked += 1
    assert checked >= 9
    rng = np.random.default_rng(3)
    for s in (0.5, 0.7, 0.8, 0.3):
        x = rng.standard_normal((3, 5, 77, 128)).astype(np.float16)
        x[0, 0, :10] = np.round(x[0, 0, :10] * 2) / 2          # ties
        got = comp.prune_magnitude(_t(x), s).cpu().numpy().view(np.uint16)
        assert np.array_equal(got, orc.prune_magnitude(x, s).view(np.uint16))


@pytest.mark.parametrize("which", ["key", "value"])
def test_compress_bit_exact(pkg, golden_dir, which):
    _, comp = pkg
    conv = comp.convert_key_batched if which == "key" else comp.convert_value_batched
    g = np.load(os.path.join(golden_dir, "compress_reference.npz"))
    names = sorted({k.split("__")[0] for k in g.files if k.endswith("__x")})
    checked = 0
    for name in names:
        x = g[f"{name}__x"]
        if x.shape[-1] != 128:
            continue
        bmp, accum, nzs = conv(_t(x.view(np.float16)))
        assert bmp.dtype == torch.int64 and accum.dtype == torch.int32 and len(nzs) == x.shape[0]
        assert np.array_equal(bmp.cpu().numpy(), g[f"{name}__{which}__bmp"]), name
        assert np.array_equal(accum.cpu().numpy(), g[f"{name}__{which}__accum"]), name
        flat = torch.cat(nzs).cpu().numpy().view(np.uint16)
        assert np.array_equal(flat, g[f"{name}__{which}__packed"]), name
        checked += 1
    assert checked >= 8
    # seeded random, larger than the fixtures, against the oracle
    for (B, t, s, seed) in [(5, 640, 0.7, 1), (2, 2048, 0.5, 2), (1, 64, 0.8, 3)]:
        c = make_cache(which, B, t, 128, s, seed)
        bmp, accum, nzs = conv(_t(c["pruned"]))
        assert np.array_equal(bmp.cpu().numpy(), c["bmp"])
        assert np.array_equal(accum.cpu().numpy(), c["idx"])
        for b in range(B):
            assert np.array_equal(nzs[b].cpu().numpy().view(np.uint16), c["nzs"][b].view(np.uint16))


@pytest.mark.parametrize("which", ["key", "value"])
def test_one_read_conversion_equals_the_two_pass_form(which):
    """mustafar_convert_onepass / _pack (round 5: the rows read once, the sizes read on the host behind that work; MUSTAFAR_CONVERT=onepass)
    against the default two-pass form, bit for bit: fixtures-sized and larger inputs, an all-zero block, rows without a single zero."""
    from mustafar_amd import compression as comp
    torch.manual_seed(3)
    cases = [torch.from_numpy(make_cache(which, B, t, 128, s, seed)["pruned"]).to(DEV) for (B, t, s, seed) in [(5, 640, 0.7, 1), (3, 2048, 0.5, 2), (1, 64, 0.8, 3)]]
    cases.append(torch.zeros((2, 128, 128), dtype=torch.float16, device=DEV))
    dense = torch.randn((2, 192, 128), device=DEV).half()
    dense[dense == 0] = 1
    cases.append(dense)
    for x in cases:
        a_bmp, a_acc, a_nz = comp._convert(x, which, onepass=False)
        b_bmp, b_acc, b_nz = comp._convert(x, which, onepass=True)
        assert torch.equal(a_bmp, b_bmp) and torch.equal(a_acc, b_acc) and len(a_nz) == len(b_nz)
        for p, q in zip(a_nz, b_nz):
            assert torch.equal(p.view(torch.int16), q.view(torch.int16))
        assert torch.cat(b_nz).data_ptr() == b_nz[0].data_ptr() if b_nz[0].numel() else True     # (pieces of one buffer: the free concatenation)
        # round 6: the sizes reach the host through a polled mirror in pinned memory (default) or a device-to-host copy: the same results
        comp._CONVERT_SYNC_COPY = True
        try:
            for form in (False, True):
                c_bmp, c_acc, c_nz = comp._convert(x, which, onepass=form)
                assert torch.equal(a_bmp, c_bmp) and torch.equal(a_acc, c_acc) and [p.numel() for p in c_nz] == [p.numel() for p in a_nz]
                assert torch.equal(torch.cat(c_nz).view(torch.int16), torch.cat(a_nz).view(torch.int16))
        finally:
            comp._CONVERT_SYNC_COPY = False
    assert comp.convert_fallbacks == 0


def test_conversion_form_by_size_and_the_mirror_reused_across_calls():
    """The default form is chosen by the number of rows (two passes below 768 Ki rows, one pass from there on: compression.py); calls of different
    head counts and forms on one thread share the pinned mirrors of the stream offsets without seeing each other's values."""
    from mustafar_amd import compression as comp
    torch.manual_seed(5)
    small = torch.from_numpy(make_cache("key", 4, 256, 128, 0.7, 11)["pruned"]).to(DEV)
    old = comp._CONVERT_ONEPASS_ROWS
    try:
        outs = []
        for rows_at in (1 << 40, 1):            # two passes, then one pass, through the PUBLIC entry point
            comp._CONVERT_ONEPASS_ROWS = rows_at
            for x in (small, small[:3], small[:, :64], small):
                bmp, acc, nz = comp.convert_key_batched(x)
                outs.append((x.shape, bmp.clone(), acc.clone(), [p.clone() for p in nz]))
        half = len(outs) // 2
        for (sa, ba, aa, na), (sb, bb, ab, nb) in zip(outs[:half], outs[half:]):
            assert sa == sb and torch.equal(ba, bb) and torch.equal(aa, ab) and all(torch.equal(p.view(torch.int16), q.view(torch.int16)) for p, q in zip(na, nb))
        assert torch.equal(outs[0][1], outs[3][1]) and torch.equal(outs[0][2], outs[3][2])
    finally:
        comp._CONVERT_ONEPASS_ROWS = old
    assert comp.convert_fallbacks == 0


def test_conversions_from_two_host_threads_on_their_own_streams():
    """The polled mirror of the stream offsets is per host THREAD (and device, head count): two threads converting different inputs on streams of their
    own, twenty times each, get what a single thread gets -- neither sees the other's sentinel or sizes."""
    import threading
    from mustafar_amd import compression as comp
    xs = [torch.from_numpy(make_cache("key", 6, 320, 128, 0.7, 31)["pruned"]).to(DEV), torch.from_numpy(make_cache("value", 6, 448, 128, 0.6, 32)["pruned"]).to(DEV)]
    fns = [comp.convert_key_batched, comp.convert_value_batched]
    want = [[t.clone() if torch.is_tensor(t) else torch.cat(t).clone() for t in fn(x)] for fn, x in zip(fns, xs)]
    torch.cuda.synchronize()
    errors = []

    def work(i):
        try:
            st = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(st):
                for _ in range(20):
                    bmp, acc, nz = fns[i](xs[i])
                    got = [bmp, acc, torch.cat(nz)]
                    st.synchronize()
                    for g, w in zip(got, want[i]):
                        if not torch.equal(g.view(torch.int16) if g.dtype == torch.float16 else g, w.view(torch.int16) if w.dtype == torch.float16 else w):
                            errors.append(f"thread {i}: mismatch")
                            return
        except Exception as e:   # noqa: BLE001 -- reported below, in the main thread
            errors.append(f"thread {i}: {e!r}")

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_empty_and_dense_blocks(pkg):
    """All-zero input (every tile empty, zero-length streams) and fully dense input (nnz = 64 everywhere)."""
    mp, comp = pkg
    z = torch.zeros((2, 128, 128), dtype=torch.float16, device=DEV)
    for conv in (comp.convert_key_batched, comp.convert_value_batched):
        bmp, accum, nzs = conv(z)
        assert not bmp.any() and not accum.any() and all(n.numel() == 0 for n in nzs)
    d = torch.randn((2, 128, 128), device=DEV).half()
    d[d == 0] = 1
    q = torch.randn((2, 1, 128), device=DEV).half()
    bmp, accum, nzs = comp.convert_key_batched(d)
    assert (bmp == -1).all() and int(accum[0, -1]) == 32 * 256
    off = torch.tensor([0, nzs[0].numel() // 8], dtype=torch.int32, device=DEV)
    out = mp.mustafar_key_formulation(bmp, torch.cat(nzs), accum, off, q, 128, 128, 2, 1)
    want = torch.einsum("btd,bnd->bnt", d.float(), q.float())
    torch.testing.assert_close(out.float(), want, rtol=2e-3, atol=2e-2)
    bmp, accum, nzs = comp.convert_value_batched(d)
    p = torch.softmax(torch.randn((2, 1, 128), device=DEV), -1).half()
    ws = torch.zeros(1, dtype=torch.float16, device=DEV)
    out = mp.mustafar_value_formulation(bmp, torch.cat(nzs), accum, off, p, ws, 128, 128, 2, 1)
    torch.testing.assert_close(out.float(), torch.einsum("bnt,btd->bnd", p.float(), d.float()), rtol=2e-3, atol=2e-3)


def test_wrapper_error_behaviour(pkg):
    """Same exceptions as mustafar_wrapper.cu:36-73 (RuntimeError), plus the added shape validation."""
    mp, _ = pkg
    c = make_cache("key", 1, 64, 128, 0.7, seed=1)
    bmp, nz, idx, off = _cache_to_dev(c)
    q = torch.zeros((1, 8, 128), dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="float16"):
        mp.mustafar_key_formulation(bmp, nz, idx, off, q.float(), 64, 128, 1, 1)
    with pytest.raises(RuntimeError, match="int64"):
        mp.mustafar_key_formulation(bmp.int(), nz, idx, off, q, 64, 128, 1, 1)
    with pytest.raises(RuntimeError, match="same device"):
        mp.mustafar_key_formulation(bmp.cpu(), nz, idx, off, q, 64, 128, 1, 1)
    with pytest.raises(RuntimeError, match="contiguous"):
        mp.mustafar_key_formulation(bmp, nz, idx, off, q.transpose(1, 2), 64, 128, 1, 1)
    with pytest.raises(RuntimeError):
        mp.mustafar_key_formulation(bmp, nz, idx, off, q, 128, 128, 1, 1)      # T does not match the cache
    with pytest.raises(RuntimeError):
        mp.mustafar_key_formulation(bmp, nz, idx, off, q, 64, 64, 1, 1)        # head_dim 64 unsupported
