"""GPU: the hot path at BASELINE.json's full sizes (one layer), checked through size-independent properties --
the CPU oracle would take minutes here.  Geometries: c1 (Llama-2-7B, 50 %, L=1024, b1: the reference's CPU-runnable case), c2 (Llama-2-7B, L=4096, b1), c3 (Llama-3-8B GQA, L=8192, b8),
c4 (L=32768, b4, 80 %), c5 (Mistral-7B geometry, L=16384, b16, 70 %).  Tolerances: integer invariants exact; SpMV vs a torch fp32 matmul over the pruned dense tensor
within fp16 (rtol 3e-3, atol scaled by sqrt(K)); linearity within the same bound."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = {"c1": (32, 32, 0.5, 1024, 1), "c2": (32, 32, 0.7, 4096, 1), "c3": (32, 8, 0.7, 8192, 8), "c4": (32, 8, 0.8, 32768, 4),
         "c5": (32, 8, 0.7, 16384, 16)}


def _popcount64(x: torch.Tensor) -> torch.Tensor:
    x = x.clone()
    cnt = torch.zeros_like(x)
    for _ in range(8):                       # 8 bytes, table-free
        b = x & 0xFF
        b = (b & 0x55) + ((b >> 1) & 0x55)
        b = (b & 0x33) + ((b >> 2) & 0x33)
        b = (b & 0x0F) + ((b >> 4) & 0x0F)
        cnt += b
        x = (x >> 8) & 0x00FFFFFFFFFFFFFF
    return cnt


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5"])
def test_full_size_invariants_and_spmv(name):
    from mustafar_amd import compression, mustafar_package as mp
    from mustafar_amd.hook import nz_offset_from_idxs
    Hq, Hkv, s, L, batch = CASES[name]
    T = ((L - 32) // 256) * 256
    Bp, BH, groups, D = batch * Hkv, batch * Hq, Hq // Hkv, 128
    gen = torch.Generator(device=DEV).manual_seed(42)
    x = torch.randn((Bp, T, D), device=DEV, generator=gen).half()
    pruned = compression.prune_magnitude(x, s)
    kth = max(1, int(s * D))
    nnz_rows = (pruned != 0).sum(-1)
    assert int(nnz_rows.min()) >= D - kth + 1 - 1            # exact zeros in the input can only lower the count by ties to 0
    assert float(nnz_rows.float().mean()) == pytest.approx(D - kth + 1, abs=0.5)
    kept = pruned != 0
    assert torch.equal(pruned[kept], x[kept])                                   # kept values are untouched
    for which in ("key", "value"):
        conv = compression.convert_key_batched if which == "key" else compression.convert_value_batched
        bmp, idx, nzs = conv(pruned)
        # --- integer invariants of the format (exact)
        assert bmp.shape == (Bp, 2 * T) and idx.shape == (Bp, 2 * T + 1)
        pop = _popcount64(bmp)
        assert int(pop.sum()) == int(kept.sum())                                  # every non-zero has its bit
        d = idx[:, 1:] - idx[:, :-1]
        assert torch.equal(d.long(), ((pop + 7) // 8) * 4)                         # ceil8(nnz)/2 per tile (compression.py:46-48)
        assert not bool(idx[:, 0].any())
        flat = torch.cat(nzs)
        assert flat.numel() == 2 * int(idx[:, -1].long().sum())
        assert int((flat != 0).sum()) == int(kept.sum())                           # padding slots are zero
        assert float(flat.float().abs().sum()) == pytest.approx(float(pruned.float().abs().sum()), rel=1e-5)  # checksum
        off = nz_offset_from_idxs(idx, Bp)
        # --- SpMV against the dense product over the pruned tensor (fp32), linearity, checksum of scores
        if which == "key":
            q1 = torch.randn((BH, 1, D), device=DEV, generator=gen).half()
            q2 = torch.randn((BH, 1, D), device=DEV, generator=gen).half()
            f = lambda q: mp.mustafar_key_formulation(bmp, flat, idx, off, q, T, D, BH, groups).float()
            Kd = pruned.float().repeat_interleave(groups, 0)                       # [BH, T, D]
            want = torch.bmm(q1.float(), Kd.transpose(1, 2))
            got = f(q1)
            torch.testing.assert_close(got, want, rtol=3e-3, atol=3e-3 * math.sqrt(D - kth + 1))
            torch.testing.assert_close(f((q1.float() + q2.float()).half()), got + f(q2), rtol=1e-2, atol=0.08)
            torch.testing.assert_close(got.sum(-1), torch.bmm(q1.float(), Kd.sum(1, keepdim=True).transpose(1, 2)).squeeze(-1),
                                       rtol=2e-2, atol=0.5 * math.sqrt(T) * 0.02 + 1.0)
        else:
            p = torch.softmax(torch.randn((BH, 1, T), device=DEV, generator=gen) * 2, -1).half()
            ws = torch.zeros(1, dtype=torch.float16, device=DEV)
            got = mp.mustafar_value_formulation(bmp, flat, idx, off, p, ws, D, T, BH, groups).float()
            Vd = pruned.float().repeat_interleave(groups, 0)
            want = torch.bmm(p.float(), Vd)
            torch.testing.assert_close(got, want, rtol=3e-3, atol=2e-4)
            for split in (1, 7):                                                   # any token split gives the same sums
                alt = mp.mustafar_value_formulation(bmp, flat, idx, off, p, ws, D, T, BH, groups, split_k=split).float()
                torch.testing.assert_close(alt, got, rtol=2e-3, atol=1e-4)
        del bmp, idx, nzs, flat
    torch.cuda.empty_cache()
