"""CPU-only: the C-ABI shared library loads and exports every function include/hbmpc_hip.h
declares, the ctypes table covers exactly that set, and the product refuses to run without a
GPU instead of falling back to anything."""
import ctypes
import os
import re

import pytest

from conftest import BLS, REPO


def declared_functions(header="hbmpc_hip.h"):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from honeybadgermpc_amd._capi import DEBUG_SYMBOLS, LIB_PATH, SYMBOLS, load_library

    headers = sorted(f for f in os.listdir(os.path.join(REPO, "include")) if f.endswith(".h"))
    assert headers == ["hbmpc_hip.h", "hbmpc_hip_debug.h"]
    names = declared_functions()
    assert len(names) >= 30
    assert sorted(SYMBOLS) == names, "ctypes table and header disagree"
    debug = declared_functions("hbmpc_hip_debug.h")
    assert sorted(DEBUG_SYMBOLS) == debug, "ctypes debug table and debug header disagree"
    assert not any(n.startswith("hb_debug") for n in names), "diagnostics do not belong in the public header"
    lib = load_library()
    raw = ctypes.CDLL(LIB_PATH)
    for name in names + debug:
        assert getattr(raw, name) is not None
    assert lib.hb_version() >= 100


def test_host_selftest_of_kernel_arithmetic():
    """the radix-2^29 Montgomery templates the kernels are built from, run on the host"""
    import random

    import numpy as np

    from honeybadgermpc_amd._capi import ints_to_limbs, limbs_to_ints, load_library, np_ptr

    lib = load_library()
    rnd = random.Random(2)
    for p, nl in [(BLS, 4), (13, 4), (53, 4), ((1 << 256) - 189, 4), ((1 << 255) - 19, 4), (13, 1), ((1 << 64) - 59, 1), (0xFFFFFFFF00000001, 1)]:
        nb = 8 * nl
        for trial in range(200):
            a, b = rnd.randrange(p), rnd.randrange(p)
            if trial == 0:
                a, b = p - 1, p - 1
            if trial == 1:
                a, b = 0, p - 1
            out = np.zeros((1, nl), dtype=np.uint64)
            rc = lib.hb_selftest_mulmod(np_ptr(ints_to_limbs([p], p + 1, nb)), nl, np_ptr(ints_to_limbs([a], p, nb)),
                                        np_ptr(ints_to_limbs([b], p, nb)), np_ptr(out))
            assert rc == 0 and limbs_to_ints(out, nb)[0] == a * b % p


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from honeybadgermpc_amd import ntl
    from honeybadgermpc_amd._capi import HbmpcBackendError

    with pytest.raises(HbmpcBackendError):
        ntl.vandermonde_batch_evaluate([1, 2], [[1, 2]], BLS)
    with pytest.raises(HbmpcBackendError):
        ntl.fft([1, 2], 5, 13, 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "honeybadgermpc_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".h", ".sh")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "hbmpc_oracle" not in src, os.path.join(root, f)
