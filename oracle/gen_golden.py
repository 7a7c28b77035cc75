#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (it reads /root/reference at run time; nothing from the
reference is copied into this repo -- only numeric inputs/outputs are saved):

  * prune_*.npz     : the reference's `dh_prune_key` / `dh_prune_value`
                      (models/llama_mustafar_kernel.py:77-153).  The module itself cannot be
                      imported (it needs the CUDA extension and transformers 4.43), so the two
                      function definitions are located with `ast` and executed with only
                      `torch` in scope, on CPU, in fp16.
  * compress_*.npz  : the reference's four Triton kernels (kernel/compression.py:8-247) run
                      under TRITON_INTERPRET=1 on CPU tensors.  The host wrappers
                      `convert_*_batched` assert `.is_cuda` (:251), so their torch glue
                      (:255-335) is restated below around the reference's own kernels.

Usage:  python oracle/gen_golden.py        (idempotent; seeds fixed)
"""
import ast
import os
import sys

os.environ["TRITON_INTERPRET"] = "1"

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# ----------------------------------------------------------------------------- prune
def load_reference_prune():
    src_path = os.path.join(REF, "models", "llama_mustafar_kernel.py")
    src = open(src_path).read()
    tree = ast.parse(src)
    fns = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in ("dh_prune_key", "dh_prune_value"):
            scope = {"torch": torch}
            exec(compile(ast.Module(body=[node], type_ignores=[]), src_path, "exec"), scope)
            fns[node.name] = scope[node.name]
    assert set(fns) == {"dh_prune_key", "dh_prune_value"}
    return fns


class _Self:
    k_sparsity = None
    v_sparsity = None


def gen_prune():
    fns = load_reference_prune()
    g = torch.Generator().manual_seed(1234)
    cases = {}
    for s in (0.5, 0.7, 0.8, 0.0, 0.99):
        x = torch.randn(1, 2, 24, 128, generator=g).to(torch.float16)
        cases[f"rand_s{s}"] = (x, s)
    # ties at the threshold: few distinct magnitudes, mixed signs
    x = (torch.randint(0, 6, (1, 1, 16, 128), generator=g).to(torch.float16) * 0.25)
    x = x * (torch.randint(0, 2, x.shape, generator=g).to(torch.float16) * 2 - 1)
    cases["ties_s0.7"] = (x, 0.7)
    cases["ties_s0.5"] = (x.clone(), 0.5)
    # zeros and negative zeros in the row; more zeros than the prune budget
    x = torch.randn(1, 1, 16, 128, generator=g).to(torch.float16)
    x[..., ::3] = 0.0
    x[..., 1::7] = -0.0
    cases["zeros_s0.7"] = (x, 0.7)
    x2 = torch.zeros(1, 1, 4, 128, dtype=torch.float16)
    x2[..., :10] = torch.randn(10, generator=g).to(torch.float16)
    cases["mostly_zero_s0.5"] = (x2, 0.5)
    # subnormals / large values
    x3 = torch.randn(1, 1, 8, 128, generator=g).to(torch.float16)
    x3[..., :32] *= 6e-5
    x3[..., 32:48] *= 1000
    cases["range_s0.8"] = (x3, 0.8)
    # head_dim 64 (format allows any D % 64 == 0)
    cases["d64_s0.7"] = (torch.randn(1, 2, 8, 64, generator=g).to(torch.float16), 0.7)

    out = {}
    for name, (x, s) in cases.items():
        yk = fns["dh_prune_key"](_Self(), x.clone(), s)
        yv = fns["dh_prune_value"](_Self(), x.clone(), s)
        assert torch.equal(yk.view(torch.int16), yv.view(torch.int16))
        out[f"{name}__x"] = x.numpy().view(np.uint16)
        out[f"{name}__y"] = yk.numpy().view(np.uint16)
        out[f"{name}__s"] = np.float64(s)
    np.savez_compressed(os.path.join(OUT, "prune_reference.npz"), **out)
    print("prune cases:", len(cases))


# ----------------------------------------------------------------------------- compression
def reference_convert(inputs: torch.Tensor, which: str):
    """Host glue of compression.py:249-339 / :341-432 on CPU around the reference's kernels."""
    sys.path.insert(0, REF)
    import kernel.compression as rc  # the reference module (Triton, interpreter mode)

    B, M, N = inputs.shape
    assert M % 64 == 0
    if which == "key":
        inputs_t = inputs.transpose(1, 2).contiguous()
        k_bitmap, k_pack = rc.calculate_bitmap_key_batched, rc.compress_key_batched
    else:
        inputs_t = inputs.contiguous()
        k_bitmap, k_pack = rc.calculate_bitmap_value_batched, rc.compress_value_batched
    tiles = (M * N) // 64
    bitmaps = torch.empty((B, tiles), dtype=torch.int64)
    counts = torch.empty((B, tiles), dtype=torch.int32)
    shifts_np = np.left_shift(np.int64(1), np.arange(63, -1, -1, dtype=np.int64))
    const_shifts = torch.tensor(shifts_np)
    grid = (tiles, B)
    stride_batch = tiles * 64
    k_bitmap[grid](inputs_t.view(-1), bitmaps.view(-1), counts.view(-1), total_elems=B * M * N,
                   shifts_ptr=const_shifts, stride_batch=stride_batch, M=M, N=N)
    accum = torch.cumsum(counts, dim=1).to(torch.int32)
    accum = torch.cat([torch.zeros((B, 1), dtype=counts.dtype), accum], dim=1).contiguous()
    total = 2 * accum[:, -1]
    offsets = torch.cumsum(total, dim=0)
    batch_offsets = torch.cat([torch.zeros(1, dtype=torch.int32), offsets[:-1]])
    packed = torch.zeros((int(offsets[-1].item()),), dtype=torch.float16)
    k_pack[grid](inputs_t.view(-1), bitmaps.view(-1), accum.view(-1), packed.view(-1), batch_offsets.view(-1),
                 total_elems=B * M * N, stride_batch=stride_batch, M=M, N=N)
    return bitmaps, accum, packed, batch_offsets.to(torch.int64)


def make_pruned(B, t, D, s, g, prune):
    x = torch.randn(1, B, t, D, generator=g).to(torch.float16)
    return prune(_Self(), x, s).reshape(B, t, D)


def gen_compress():
    prune = load_reference_prune()["dh_prune_key"]
    g = torch.Generator().manual_seed(4321)
    cases = {}
    for (B, t, s) in [(1, 64, 0.7), (2, 64, 0.5), (3, 64, 0.8), (2, 256, 0.7), (1, 512, 0.7), (2, 128, 0.0)]:
        cases[f"rand_B{B}_t{t}_s{s}"] = make_pruned(B, t, 128, s, g, prune)
    # edge tiles: all-zero tiles, full tiles (nnz = 64), nnz multiple of 8, -0.0 entries
    x = torch.zeros(2, 128, 128, dtype=torch.float16)
    dense = torch.randn(2, 128, 128, generator=g).to(torch.float16)
    dense[dense == 0] = 1.0
    x[0, :64, :] = dense[0, :64, :]                 # K: full tiles; V: full tiles
    x[0, 64:, 5] = dense[0, 64:, 5]                 # K: one full tile among empty ones
    x[1, 3, :] = dense[1, 3, :]                     # V: one token fully dense, rest empty
    x[1, 64:72, 64:72] = dense[1, 64:72, 64:72]     # nnz = 8 per touched tile
    x[1, 100, 0:16] = dense[1, 100, 0:16]           # V tile nnz = 16
    x[1, 80:96, 127] = dense[1, 80:96, 127]         # K tile nnz = 16
    x[1, 120, 64:] = -0.0                           # negative zeros must not set bits
    cases["edge_tiles"] = x
    # adversarial: every kept value in one 64-channel half
    y = torch.zeros(1, 64, 128, dtype=torch.float16)
    y[0, :, :40] = dense[0, :64, :40]
    cases["one_half"] = y
    # D = 64
    cases["d64"] = make_pruned(2, 64, 64, 0.7, g, prune)

    out = {}
    for name, x in cases.items():
        for which in ("key", "value"):
            bmp, accum, packed, boffs = reference_convert(x.clone(), which)
            out[f"{name}__{which}__bmp"] = bmp.numpy()
            out[f"{name}__{which}__accum"] = accum.numpy()
            out[f"{name}__{which}__packed"] = packed.numpy().view(np.uint16)
            out[f"{name}__{which}__batch_offsets"] = boffs.numpy()
        out[f"{name}__x"] = x.numpy().view(np.uint16)
        print("compress case", name, tuple(x.shape))
    np.savez_compressed(os.path.join(OUT, "compress_reference.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_prune()
    gen_compress()
    print("golden vectors written to", os.path.abspath(OUT))
