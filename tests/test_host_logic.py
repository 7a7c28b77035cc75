"""CPU: host-side logic of the package (no kernels run): C-ABI surface, wrapper argument checks, cache append."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mustafar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int|int64_t)\s+(\w+)\s*\(", text, flags=re.M)))


def test_header_declares_the_reference_entry_points():
    syms = _declared_symbols()
    for s in ("Key_SplitK_API", "Value_SplitK_API", "mustafar_prune_magnitude", "mustafar_compress_bitmap_key",
              "mustafar_compress_bitmap_value", "mustafar_compress_pack_key", "mustafar_compress_pack_value"):
        assert s in syms


def test_cabi_library_exports_every_declared_symbol():
    """The C-ABI library loads here (no GPU) and exports every symbol include/mustafar_hip.h declares."""
    from mustafar_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python __graft_entry__.py` (build) first"
    L = _lib.load()
    syms = _declared_symbols()
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} not exported"
    assert L.mustafar_abi_version() >= 106
    # pure host helpers (no device access)
    s = L.mustafar_value_pick_split_k(128, 1, 7936, 256, 4)
    assert 1 <= s <= 31
    assert L.mustafar_value_workspace_bytes(128, 1, 7936, 256, 4, 1) == 0
    assert L.mustafar_value_workspace_bytes(128, 1, 7936, 256, 4, s) >= s * 256 * 128 * 4
    assert L.mustafar_value_pick_split_k(128, 1, 64, 1, 1) == 1


def test_cache_view_strides_are_validated_without_a_gpu():
    """mustafar_decode_attention_view rejects head strides shorter than the tokens in use (host check, no launch)."""
    from mustafar_amd import _lib
    L = _lib.load()
    one = 8   # never dereferenced: validation fails first
    T = 128
    ok_args = (one, one, one, None, None, 1, 64, one, T + 64, one, one, 1, T, 4, 1, ctypes.c_float(11.3), None, None, 0, 0, 0)
    short = _lib.CacheView(one, one, one, one, 2 * T - 1, 0)
    good = _lib.CacheView(one, one, one, one, 2 * T, 2 * T + 1)
    assert L.mustafar_decode_attention_view(None, ctypes.byref(short), ctypes.byref(good), *ok_args) == 1
    assert L.mustafar_decode_attention_view(None, ctypes.byref(good), None, *ok_args) == 1
    short_idx = _lib.CacheView(one, one, one, one, 0, 2 * T)
    assert L.mustafar_decode_attention_view(None, ctypes.byref(good), ctypes.byref(short_idx), *ok_args) == 1
    # undefined bits of `flags` (engine field 4..7, structure field 3, anything above bit 5) are rejected before any launch
    for bad in (4, 7, 3 << 4, 1 << 6, 1 << 31):
        assert L.mustafar_decode_attention_view(None, ctypes.byref(good), ctypes.byref(good), *(ok_args[:-1] + (bad,))) == 1


def test_extent_arguments_are_validated_without_a_gpu():
    """mustafar_decode_attention_extents: T_base and T - T_base are multiples of 256, the tables are required once T > T_base, the
    base views need a stream stride; mustafar_decode_reads_extents answers per GQA shape / row pitch / flags (host only)."""
    from mustafar_amd import _lib
    L = _lib.load()
    one = 8   # never dereferenced: validation fails first
    view = _lib.CacheView(one, one, one, one, 2 * 512, 2 * 512 + 1, 4096)
    nostride = _lib.CacheView(one, one, one, one, 2 * 512, 2 * 512 + 1, 0)
    def call(kv, vv, T_base, kt, vt, T, t_dev=None):
        tail = (one, one, one, None, None, 1, 64, one, T + 64, one, one, 1, T, 8, 4, ctypes.c_float(11.3), None, None, 0, 0, 0, t_dev)
        return L.mustafar_decode_attention_extents(None, ctypes.byref(kv) if kv else None, ctypes.byref(vv) if vv else None, T_base, kt, vt, *tail)
    assert call(view, view, 500, one, one, 756) == 1          # T_base % 256
    assert call(view, view, 512, one, one, 512 + 128) == 1    # (T - T_base) % 256
    assert call(view, view, 512, one, one, 256) == 1          # T < T_base
    assert call(view, view, 512, None, one, 768) == 1         # tables missing
    assert call(nostride, view, 512, one, one, 768) == 1      # base views without a stream stride
    assert call(None, view, 512, one, one, 768) == 1
    assert call(view, view, 512, one, one, 512, t_dev=one) == 1   # a device-side T needs a capacity beyond the base tokens
    assert L.mustafar_decode_reads_extents(4, 8256, 0) == 1 and L.mustafar_decode_reads_extents(8, 8256, 0) == 1
    assert L.mustafar_decode_reads_extents(1, 8256, 0) == 1 and L.mustafar_decode_reads_extents(2, 8256, 0) == 1    # MHA, GQA-2: the pair form since round 4
    assert L.mustafar_decode_reads_extents(4, 32 * 5000, 0) == 0                                                 # rows beyond the row kernel's slab count
    assert L.mustafar_decode_reads_extents(4, 8256 + 8, 0) == 0                                                  # row pitch % 32
    assert L.mustafar_decode_reads_extents(4, 8256, 1 << 4) == 0 and L.mustafar_decode_reads_extents(4, 8256, 2 << 4) == 1   # two launches / one-pass asked for
    assert L.mustafar_decode_reads_extents(4, 8256, 1 << 6) == 0                                                 # undefined flag bits


def test_batched_trigger_arguments_are_validated_without_a_gpu():
    """mustafar_trigger_compress_batch / mustafar_trigger_finish_batch (round 4): every item is checked on the host before anything is
    launched -- null windows, views without room for the 256 tokens, a missing flag, shapes."""
    from mustafar_amd import _lib
    L = _lib.load()
    one = 8   # never dereferenced: validation fails first
    good = _lib.CacheView(one, one, one, one, 2 * 256, 2 * 256 + 1, 4096)
    short = _lib.CacheView(one, one, one, one, 2 * 192, 2 * 192 + 1, 4096)
    def items(**kw):
        arr = (_lib.TriggerItem * 2)()
        for it in arr:
            it.k_window = it.v_window = one
            it.k_dst, it.v_dst = good, good
            it.k_head_total = it.v_head_total = it.overflow_flag = one
        for k, v in kw.items():
            setattr(arr[1], k, v)
        return arr
    args = (288 * 128, 8, 256, 128, 89, 89, 32768, 32768, one)
    assert L.mustafar_trigger_compress_batch(None, 0, items(), *args) == 1                       # no items
    assert L.mustafar_trigger_compress_batch(None, 2, None, *args) == 1
    assert L.mustafar_trigger_compress_batch(None, 2, items(k_window=None), *args) == 1          # second item: null window
    assert L.mustafar_trigger_compress_batch(None, 2, items(overflow_flag=None), *args) == 1     # the flag is required
    assert L.mustafar_trigger_compress_batch(None, 2, items(v_dst=short), *args) == 1            # rows too short for 256 tokens
    assert L.mustafar_trigger_compress_batch(None, 2, items(), 100 * 128, *args[1:]) == 1        # window rows shorter than t
    assert L.mustafar_trigger_compress_batch(None, 2, items(), *(args[:2] + (200,) + args[3:])) == 1   # t % 64
    assert L.mustafar_trigger_compress_batch(None, 2, items(), *(args[:-1] + (None,))) == 1      # no scratch
    assert L.mustafar_trigger_finish_batch(None, 2, items(v_window=None), 288 * 128, 8, 288, 256) == 1
    assert L.mustafar_trigger_finish_batch(None, 2, items(), 288 * 128, 8, 200, 256) == 1        # len < drop
    assert L.mustafar_trigger_finish_batch(None, 2, items(), 288 * 128, 8, 400, 256) == 1        # rows beyond the head stride


def test_invalid_arguments_are_rejected_without_a_gpu():
    """Shape errors are caught on the host before any launch (MUSTAFAR_EINVAL == 1)."""
    from mustafar_amd import _lib
    L = _lib.load()
    one = ctypes.c_void_p(8)   # never dereferenced: validation fails first
    assert L.Key_SplitK_API(None, None, one, one, one, one, one, one, 100, 8, 128, None, 1, 4, 1) == 1     # T % 64
    assert L.Key_SplitK_API(None, None, one, one, one, one, one, one, 128, 8, 64, None, 1, 4, 1) == 1      # head_dim
    assert L.Key_SplitK_API(None, None, one, one, one, one, one, one, 128, 3, 128, None, 1, 4, 1) == 1     # N
    assert L.Key_SplitK_API(None, None, one, one, one, one, one, one, 128, 8, 128, None, 1, 6, 4) == 1     # batch % groups
    assert L.Key_SplitK_API(None, None, None, one, one, one, one, one, 128, 8, 128, None, 1, 4, 1) == 1    # null
    assert L.Value_SplitK_API(None, None, one, one, one, one, one, one, 64, 8, 128, None, 1, 4, 1) == 1    # M != 128
    assert L.Value_SplitK_API(None, None, one, one, one, one, one, one, 128, 8, 256, None, 4, 4, 1) == 1   # ws missing
    assert L.mustafar_prune_magnitude(None, one, one, 4, 64, 10) == 1
    assert L.mustafar_prune_magnitude(None, one, one, 4, 128, 0) == 1
    assert L.mustafar_compress_bitmap_key(None, one, 2, 100, 128, one, one, one) == 1


def test_wrapper_checks_match_reference_messages():
    """mustafar_wrapper.cu:36-73: dtype/device errors are RuntimeErrors with the reference's messages."""
    from mustafar_amd import mustafar_package as mp
    bmp = torch.zeros(4, dtype=torch.int64)
    nz = torch.zeros(8, dtype=torch.float16)
    idx = torch.zeros(5, dtype=torch.int32)
    off = torch.zeros(1, dtype=torch.int32)
    B = torch.zeros((1, 8, 128), dtype=torch.float16)
    with pytest.raises(RuntimeError, match="Tensor B must be of type float16"):
        mp.mustafar_key_formulation(bmp, nz, idx, off, B.float(), 64, 128, 1, 1)
    with pytest.raises(RuntimeError, match="Tensor NZ must be of type float16"):
        mp.mustafar_key_formulation(bmp, nz.float(), idx, off, B, 64, 128, 1, 1)
    with pytest.raises(RuntimeError, match="Tensor bmp must be of type int64"):
        mp.mustafar_key_formulation(bmp.int(), nz, idx, off, B, 64, 128, 1, 1)
    with pytest.raises(RuntimeError, match="Tensor idx must be of type int"):
        mp.mustafar_value_formulation(bmp, nz, idx.long(), off, B, nz, 128, 64, 1, 1)
    with pytest.raises(RuntimeError, match="Tensor NZ_Offset must be of type int"):
        mp.mustafar_value_formulation(bmp, nz, idx, off.long(), B, nz, 128, 64, 1, 1)
    with pytest.raises(RuntimeError, match="must be on CUDA device"):   # all CPU: the reference's TORCH_CHECK (:70-73)
        mp.mustafar_key_formulation(bmp, nz, idx, off, B, 64, 128, 1, 1)


def test_kth_matches_reference_expression():
    from mustafar_amd.compression import kth_from_sparsity
    for s, want in ((0.5, 64), (0.7, 89), (0.8, 102), (0.0, 1), (0.99, 126), (0.3, 38)):
        assert kth_from_sparsity(s, 128) == want == orc.kth_from_sparsity(s, 128)


@pytest.mark.parametrize("which", ["key", "value"])
def test_cache_append_equals_one_shot_compression(which):
    """Property of the format + hook append (model :339-390): compress(A) ++ compress(B) == compress(A||B)."""
    from mustafar_amd.hook import FlatStreams, append_compressed, nz_offset_from_idxs
    rng = np.random.default_rng(0)
    heads, D = 3, 128
    x = orc.prune_magnitude(rng.standard_normal((heads, 768, D)).astype(np.float16), 0.7)
    conv = orc.convert_key_batched if which == "key" else orc.convert_value_batched

    def pack(xs):
        bmp, accum, nzs = conv(xs)
        idxs = torch.from_numpy(accum)
        return [torch.from_numpy(bmp), idxs, FlatStreams([torch.from_numpy(n) for n in nzs]), nz_offset_from_idxs(idxs, heads)]

    cache = pack(x[:, :256])
    cache = append_compressed(cache, pack(x[:, 256:512]), heads, 256, 256, D)
    cache = append_compressed(cache, pack(x[:, 512:768]), heads, 512, 256, D)
    want = pack(x)
    assert torch.equal(cache[0].view(heads, -1), want[0])
    assert torch.equal(cache[1].view(heads, -1), want[1])
    assert torch.equal(cache[3], want[3])
    assert torch.equal(torch.cat(cache[2]).view(torch.int16), torch.cat(want[2]).view(torch.int16))
    assert torch.equal(cache[2].flat.view(torch.int16), want[2].flat.view(torch.int16))
    assert np.array_equal(cache[3].numpy(), orc.nz_offset_from_idx(want[1].numpy()))
    # the appended cache drives the oracle SpMV to the same scores as the one-shot cache
    q = rng.standard_normal((heads, 1, D)).astype(np.float16)
    if which == "key":
        a, _ = orc.key_spmv(cache[0].numpy(), cache[2].flat.numpy(), cache[1].numpy(), cache[3].numpy(), q, 768, D, heads, 1)
        b, _ = orc.key_spmv(want[0].numpy(), want[2].flat.numpy(), want[1].numpy(), want[3].numpy(), q, 768, D, heads, 1)
        assert np.array_equal(a.view(np.uint16), b.view(np.uint16))


def test_compiled_extension_exposes_reference_module_surface():
    """kernel/kernel_wrapper/pybind.cpp:7-10: module `mustafar_package` with exactly these two functions."""
    import sys
    dropin = os.path.join(ROOT, "mustafar_amd", "dropin")
    sys.path.insert(0, dropin)
    try:
        import mustafar_package
        import kernel.compression as compression
    finally:
        sys.path.remove(dropin)
    assert mustafar_package.__file__.endswith(".so")
    assert callable(mustafar_package.mustafar_key_formulation) and callable(mustafar_package.mustafar_value_formulation)
    assert callable(compression.convert_key_batched) and callable(compression.convert_value_batched)
    bmp = torch.zeros(4, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="Tensor B must be of type float16"):
        mustafar_package.mustafar_key_formulation(bmp, torch.zeros(8, dtype=torch.float16), torch.zeros(5, dtype=torch.int32),
                                                  torch.zeros(1, dtype=torch.int32), torch.zeros((1, 8, 128)), 64, 128, 1, 1)


def test_no_read_of_an_in_flight_scalar_load_in_the_spmv_isa():
    """The asm inner loop issues scalar loads and waits later; the generated ISA must not read (copy, spill) one of
    those SGPRs in between -- there is no hardware interlock (tools/check_smem_hazards.py, compiles device code only)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_smem_hazards", os.path.join(ROOT, "tools", "check_smem_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0


def test_build_variant_refuses_an_isa_that_fails_the_hazard_checks():
    """tools/build_variant.sh (round 6): a probe / experiment library is only emitted when its ISA passes checks (1)-(4) of
    tools/check_smem_hazards.py.  One allowed variant (a row-kernel knob: the ISA stays clean, the library is written) and one refused
    one (-DMUSTAFAR_META_EARLY=2: the compiler spills scalar registers that loads are still writing -- DESIGN 4.1 item 6 -- exit code 3,
    no library), built side by side; the knob of round 5's faulting probe does not compile at all any more."""
    import subprocess
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    script = os.path.join(ROOT, "tools", "build_variant.sh")
    vdir = os.path.join(ROOT, "mustafar_amd", "lib", "variants")
    names = {"ok": "citest_allowed", "bad": "citest_refused", "gone": "citest_nometawait"}
    for n in names.values():
        try:
            os.remove(os.path.join(vdir, f"libmustafar_hip_{n}.so"))
        except OSError:
            pass
    procs = {"ok": subprocess.Popen(["bash", script, names["ok"], "-DMUSTAFAR_FINISH_EARLY=16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True),
             "bad": subprocess.Popen(["bash", script, names["bad"], "-DMUSTAFAR_META_EARLY=2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True),
             "gone": subprocess.Popen(["bash", script, names["gone"], "-DMUSTAFAR_PROBE_NOMETAWAIT"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)}
    out = {k: p.communicate(timeout=900) for k, p in procs.items()}
    try:
        assert procs["ok"].returncode == 0, out["ok"][1][-2000:]
        assert os.path.exists(os.path.join(vdir, f"libmustafar_hip_{names['ok']}.so"))
        assert procs["bad"].returncode == 3, (procs["bad"].returncode, out["bad"][1][-2000:])
        assert "REFUSED" in out["bad"][1] and "hazards:" in out["bad"][1]
        assert not os.path.exists(os.path.join(vdir, f"libmustafar_hip_{names['bad']}.so"))
        assert procs["gone"].returncode not in (0, 3) and "MUSTAFAR_PROBE_NOMETAWAIT was removed" in out["gone"][1]
        assert not os.path.exists(os.path.join(vdir, f"libmustafar_hip_{names['gone']}.so"))
    finally:
        for n in names.values():
            try:
                os.remove(os.path.join(vdir, f"libmustafar_hip_{n}.so"))
            except OSError:
                pass


def test_stream_pieces_make_the_hooks_torch_cat_free():
    """compression.StreamPiece (CPU tensors suffice): `torch.cat` of all the pieces of one buffer, in order, IS that buffer (model
    :274, :314); the trigger's per-head `torch.cat([old[b], new[b]])` (model :368, :390) fills one new buffer whose pieces
    concatenate for free again; anything else falls back to an ordinary copy with ordinary results."""
    from mustafar_amd.compression import StreamPiece, pieces_of
    flat = torch.arange(40, dtype=torch.float16)
    offs = [0, 8, 8, 24, 40]                                   # head 1 is empty
    ps = pieces_of(flat, offs)
    assert all(type(p) is StreamPiece for p in ps) and [p.numel() for p in ps] == [8, 0, 16, 16]
    whole = torch.cat(ps)
    assert type(whole) is torch.Tensor and whole.data_ptr() == flat.data_ptr() and whole.numel() == 40
    assert torch.cat(ps, dim=0).data_ptr() == flat.data_ptr() and torch.cat(tuple(ps)).data_ptr() == flat.data_ptr()
    # not the whole buffer / not in order / mixed with a plain tensor: a copy, with the values torch.cat always gives
    part = torch.cat(ps[:3])
    assert part.data_ptr() != flat.data_ptr() and torch.equal(part, flat[:24])
    perm = torch.cat([ps[2], ps[0], ps[1], ps[3]])
    assert perm.data_ptr() != flat.data_ptr() and torch.equal(perm, torch.cat([flat[8:24], flat[:8], flat[24:]]))
    mixed = torch.cat([ps[0], torch.ones(3, dtype=torch.float16)])
    assert type(mixed) is torch.Tensor and mixed.tolist() == list(range(8)) + [1, 1, 1]
    # the trigger: new tokens of every head in their own buffer, concatenated per head exactly as the model writes it
    new = pieces_of(torch.arange(100, 124, dtype=torch.float16), [0, 8, 16, 16, 24])
    merged = [torch.cat([ps[b], new[b]], dim=0) for b in range(4)]
    assert all(type(m) is StreamPiece for m in merged)
    for b in range(4):
        assert torch.equal(merged[b].as_subclass(torch.Tensor), torch.cat([flat[offs[b]:offs[b + 1]], new[b].as_subclass(torch.Tensor)]))
    again = torch.cat(merged)
    assert again.data_ptr() == merged[0].data_ptr() and again.numel() == 64
    assert torch.equal(again, torch.cat([m.as_subclass(torch.Tensor).clone() for m in merged]))
    # the same per-head concatenation asked for a SECOND time: a fresh tensor (the slot of the shared buffer belongs to the first
    # result, which aliases the cache -- INTEGRATION.md "Aliasing"); an in-place op on it leaves the first result alone
    twice = torch.cat([ps[0], new[0]], dim=0)
    assert twice.data_ptr() != merged[0].data_ptr() and torch.equal(twice.as_subclass(torch.Tensor), merged[0].as_subclass(torch.Tensor))
    twice.as_subclass(torch.Tensor).zero_()
    assert float(merged[0][0]) == 0.0 and float(merged[0][1]) == 1.0 and float(again[1]) == 1.0
    # ordinary tensor behaviour of a piece
    assert float(ps[0][-1]) == 7.0 and type(ps[2] * 2) is torch.Tensor and ps[3].view(2, 8).shape == (2, 8) and len(ps[2]) == 16


def _checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_smem_hazards", os.path.join(ROOT, "tools", "check_smem_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_FAKE_KERNEL = """\t.text
_Zfake_kernel:
\ts_load_dwordx4 s[4:7], s[0:1], 0x0
\ts_waitcnt lgkmcnt(0)
%s
\ts_endpgm
.Lfunc_end0:
"""


def test_checker_flags_a_dot_result_without_its_wait_states():
    """tools/check_smem_hazards.py check (6), on hand-written ISA: a v_dot2 at the end of an asm statement (the round-6 bug: the compiler put a
    move of the accumulator one wait state behind it), the same with `s_nop 3` behind it, and a different vector instruction inside the
    statement that reads / overwrites the result too early."""
    mod = _checker()
    dots = lambda body: mod.check_dot_hazards(_FAKE_KERNEL % body)
    bad, n = dots("\t;;#ASMSTART\n\tv_dot2_f32_f16 v9, v23, s4, v9\n\t;;#ASMEND\n\tv_mov_b32_e32 v29, v9")
    assert n == 1 and len(bad) == 1 and "wait state" in bad[0][2]
    bad, n = dots("\t;;#ASMSTART\n\tv_dot2_f32_f16 v9, v23, s4, v9\n\ts_nop 3\n\t;;#ASMEND\n\tv_mov_b32_e32 v29, v9")
    assert n == 1 and not bad
    # back-to-back accumulation by the same opcode is fine; four independent dots behind the last one are its wait states
    bad, n = dots("\t;;#ASMSTART\n\tv_dot2_f32_f16 v9, v23, s4, v9\n\tv_dot2_f32_f16 v9, v24, s5, v9\n\tv_dot2_f32_f16 v6, v23, s4, v6\n"
                  "\tv_dot2_f32_f16 v7, v23, s4, v7\n\tv_dot2_f32_f16 v8, v23, s4, v8\n\tv_dot2_f32_f16 v5, v23, s4, v5\n\ts_nop 3\n\t;;#ASMEND")
    assert n == 6 and not bad
    bad, _ = dots("\t;;#ASMSTART\n\tv_dot2_f32_f16 v9, v23, s4, v9\n\tv_add_f32_e32 v1, v9, v9\n\ts_nop 3\n\t;;#ASMEND")      # read 0 wait states behind
    assert len(bad) == 1 and "touches the result" in bad[0][2]
    bad, _ = dots("\t;;#ASMSTART\n\tv_dot2_f32_f16 v9, v23, s4, v9\n\ts_nop 2\n\tv_mov_b32 v9, 0\n\ts_nop 3\n\t;;#ASMEND")          # overwritten 3 behind: needs 4
    assert len(bad) == 1


def test_checker_tracks_divergent_regions_by_the_saved_mask():
    """Check (2): an asm statement that loads EXEC with a bitmap may not sit inside a compiler-made divergent region.  An inner `if` at the very end of an
    outer one is closed by the OUTER restore alone (round 6: counting restores against saves left the depth at 1 for the rest of the kernel)."""
    mod = _checker()
    asm = "\t;;#ASMSTART\n\ts_mov_b64 exec, s[20:21]\n\tv_mov_b32 v1, 0\n\ts_mov_b64 exec, -1\n\t;;#ASMEND"
    run = lambda body: [h for h in mod.check(_FAKE_KERNEL % body)[0] if "divergent" in h[2]]
    nested = "\ts_and_saveexec_b64 s[8:9], vcc\n\ts_and_saveexec_b64 s[10:11], vcc\n\ts_xor_b64 s[10:11], exec, s[10:11]\n\tv_mov_b32_e32 v2, 0\n\ts_or_b64 exec, exec, s[8:9]\n"
    assert not run(nested + asm), "both regions are closed by the outer restore"
    assert len(run("\ts_and_saveexec_b64 s[8:9], vcc\n" + asm + "\n\ts_or_b64 exec, exec, s[8:9]")) == 1, "an asm statement inside an open region is reported"
    assert not run("\ts_and_saveexec_b64 s[8:9], vcc\n\tv_mov_b32_e32 v2, 0\n\ts_or_b64 exec, exec, s[8:9]\n" + asm)
    # a ROTATED loop (round 6): the exit label -- and with it the restore -- stands ABOVE the back edge that sheds the lanes; nothing below it is inside
    rotated = (".LBB0_4:\n\ts_or_b64 exec, exec, s[0:1]\n\ts_branch .LBB0_9\n.LBB0_5:\n\tv_mov_b32_e32 v2, 0\n\ts_andn2_b64 exec, exec, s[2:3]\n"
               "\ts_cbranch_execz .LBB0_4\n\ts_branch .LBB0_5\n.LBB0_9:\n")
    assert not run(rotated + asm), "the loop's lanes come back at the label above"
    rotated2 = (".LBB0_4:\n\ts_or_b64 exec, exec, s[0:1]\n\ts_branch .LBB0_9\n.LBB0_5:\n\tv_mov_b32_e32 v2, 0\n\ts_andn2_b64 exec, exec, s[2:3]\n"
                "\ts_cbranch_execnz .LBB0_5\n\ts_branch .LBB0_4\n.LBB0_9:\n")
    assert not run(rotated2 + asm), "the same loop written as `continue while lanes remain`, then a branch up to the restore"
    # ... and the last `if` of such a loop body: saved below, restored at the loop's head above; its region is the block that follows, up to the next label
    tail_if = ".LBB0_4:\n\ts_or_b64 exec, exec, s[8:9]\n\tv_mov_b32_e32 v2, 0\n\ts_and_saveexec_b64 s[8:9], vcc\n\ts_cbranch_execz .LBB0_4\n"
    assert len(run(tail_if + asm + "\n\ts_branch .LBB0_4\n.LBB0_9:\n")) == 1, "inside the block behind the save"
    assert not run(tail_if + "\tv_mov_b32_e32 v3, 0\n\ts_branch .LBB0_4\n.LBB0_9:\n" + asm), "behind the next label the region is over"


def test_the_polled_mirror_of_the_stream_offsets_waits_and_gives_up_correctly():
    """compression._await_offsets (round 6; no GPU: a numpy array stands for the pinned mirror, an object with query() for the stream): returns once no
    entry is the negative sentinel, whatever order the entries arrive in; a stream that has drained while entries are still negative is an error, not a wait."""
    import threading
    import time
    from mustafar_amd import compression as comp

    class Stream:
        def __init__(self, done):
            self.done, self.queries = done, 0

        def query(self):
            self.queries += 1
            return self.done

    arr = np.full(9, -1, np.int64)
    want = [0, 8, 8, 24, 40, 40, 64, 72, 96]

    def device():   # the entries land one by one, back to front
        for i in reversed(range(9)):
            time.sleep(0.002)
            arr[i] = want[i]

    t = threading.Thread(target=device)
    t.start()
    got = comp._await_offsets(arr, Stream(False))
    t.join()
    assert got == want
    # nothing ever arrives and the stream reports itself drained: raise instead of spinning for ever
    arr[:] = -1
    st = Stream(True)
    with pytest.raises(RuntimeError, match="no stream offsets"):
        comp._await_offsets(arr, st)
    assert st.queries >= 1
    # everything already there: no query at all
    arr[:] = want
    st = Stream(False)
    assert comp._await_offsets(arr, st) == want and st.queries == 0
