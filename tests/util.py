"""Shared helpers for the parity tests (CPU oracle side)."""
import numpy as np

from oracle import oracle as orc


def make_cache(which, B, t, D, sparsity, seed, adversarial=False):
    """Random N(0,1) fp16 block -> reference prune rule -> reference format (all via the oracle)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, t, D)).astype(np.float16)
    if adversarial:   # every kept value in one 64-channel half: nnz = 0 and nnz >= 56 tiles
        x[:, :, 64:] *= np.float16(1e-3)
    xp = orc.prune_magnitude(x, sparsity)
    conv = orc.convert_key_batched if which == "key" else orc.convert_value_batched
    bmp, accum, nzs = conv(xp)
    return dict(x=x, pruned=xp, bmp=bmp, idx=accum, nzs=nzs, nz_offset=orc.nz_offset_from_idx(accum))


def fp16_bound(ref64, sumabs):
    """|fp16(fp32-accumulated sum) - exact| for any summation order: one fp16 rounding of the result
    (2^-11 relative, doubled for slack) + fp32 accumulation noise proportional to sum|terms| + fp16 tiny."""
    return 2.0 ** -10 * np.abs(ref64) + 4e-6 * sumabs + 1e-7
