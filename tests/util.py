"""Shared helpers for the parity tests (CPU oracle side)."""
import math

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


DENSE_ULPS = 3.0    # fused vs fp32 dense attention: fp16 ulps of the output scale (the scores are rounded to fp16 twice on the way, model :278, :284)
NATIVE_ULPS = 2.0   # fused vs the unfused call sequence (same score roundings; they differ in where the probabilities are normalised)


def excess(got, want, ulps):
    """max |got - want| in units of the bound `ulps` x 2^-11 x max|want| + 1e-4 (> 1: outside).  Relative to the output SCALE, so that
    one lost 64-token block shows at every cache length (an absolute atol of 2e-3 is the size of the outputs themselves at 32 k tokens)."""
    w = want.float()
    scale = max(float(w.abs().max()), 2.0 ** -6)
    err = float((got.float() - w).abs().max())
    return err / (ulps * 2.0 ** -11 * scale + 1e-4) if math.isfinite(err) else float("inf")
