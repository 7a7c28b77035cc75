"""CPU: the bit-sliced k-th-smallest-magnitude search of the compression kernel (mustafar_amd/csrc/select_kth.h, shared by the
device code and this host build) against numpy on random, tie-heavy and degenerate rows; k as the prune rule makes it
(models/llama_mustafar_kernel.py:88-103: k = max(1, int(s * 128)), kthvalue = k-th smallest magnitude)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("selkth") / "libselect_kth_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "csrc", "select_kth_host.cpp"), "-o", out])
    L = ctypes.CDLL(out)
    L.kth_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.transpose_block.argtypes = [ctypes.c_void_p]
    return L


def test_transpose_orientation(lib):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 2 ** 32, 32, dtype=np.uint64).astype(np.uint32)
    t = a.copy()
    lib.transpose_block(t.ctypes.data)
    for b in range(32):
        for i in range(32):
            assert (int(t[31 - b]) >> (31 - i)) & 1 == (int(a[i]) >> b) & 1


def _rows():
    rng = np.random.default_rng(1)
    rows = [rng.standard_normal((400, 128)).astype(np.float16)]
    rows.append((rng.integers(-3, 4, (200, 128)) * 0.5).astype(np.float16))            # heavy ties, zeros, both signs
    rows.append(np.zeros((3, 128), np.float16))
    rows.append(np.full((2, 128), -0.0, np.float16))
    rows.append((rng.standard_normal((50, 128)) * 1e-6).astype(np.float16))             # subnormals
    big = rng.standard_normal((50, 128)).astype(np.float32) * 3e4
    rows.append(np.clip(big, -65504, 65504).astype(np.float16))                         # up to the largest finite magnitude
    one = np.zeros((4, 128), np.float16); one[:, 77] = 5.0
    rows.append(one)
    return np.concatenate(rows)


@pytest.mark.parametrize("kth", [1, 2, 38, 64, 89, 102, 126, 127, 128])
def test_kth_magnitude_matches_numpy(lib, kth):
    x = _rows()
    mags = (x.view(np.uint16) & 0x7fff)
    want = np.sort(mags, axis=1)[:, kth - 1].astype(np.uint32)
    words = np.ascontiguousarray(x.view(np.uint16)).view(np.uint32).reshape(x.shape[0], 64)   # element 2j low, 2j + 1 high
    got = np.empty(x.shape[0], np.uint32)
    lib.kth_rows(words.ctypes.data, x.shape[0], kth, got.ctypes.data)
    assert np.array_equal(got, want)
