import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
BLS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _unstr(v):
    """inverse of gen_golden.S: decimal strings -> ints, recursively"""
    if isinstance(v, str):
        try:
            return int(v)
        except ValueError:
            return v
    if isinstance(v, list):
        return [_unstr(x) for x in v]
    if isinstance(v, dict):
        return {k: _unstr(x) for k, x in v.items()}
    return v


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return _unstr(json.load(f))


@pytest.fixture(scope="session")
def golden():
    return load_golden


_NTL_NAMES = (
    "lagrange_interpolate", "evaluate", "vandermonde_batch_interpolate", "vandermonde_batch_evaluate",
    "fft", "partial_fft", "fft_batch_evaluate", "fft_interpolate", "fft_batch_interpolate", "gao_interpolate", "gao_interpolate_batch",
    "vandermonde_inverse", "sqrt_mod",
)


def install_oracle_backend(monkeypatch):
    """TEST-ONLY: route the package's arithmetic entry points to the CPU oracle so that the
    host logic (codec classes, IncrementalDecoder, batch_reconstruct) can be exercised
    without a GPU.  The product never does this."""
    import oracle
    import honeybadgermpc_amd.device as dev
    import honeybadgermpc_amd.ntl as ntl
    import honeybadgermpc_amd.polynomial as poly
    import honeybadgermpc_amd.reed_solomon as rs

    for name in _NTL_NAMES:
        monkeypatch.setattr(ntl, name, getattr(oracle, name))
        if hasattr(rs, name):
            monkeypatch.setattr(rs, name, getattr(oracle, name))
    monkeypatch.setattr(poly, "fft_cpp", oracle.fft)
    monkeypatch.setattr(poly, "fft_interpolate_cpp", oracle.fft_interpolate)
    monkeypatch.setattr(dev, "wb_decode_batch", oracle.wb_decode_batch)

    def _ie(*a, **k):
        raise AssertionError("unreachable")

    # the oracle raises its own InterpolationError class; make the package's name point at it
    monkeypatch.setattr(ntl, "InterpolationError", oracle.InterpolationError)


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    """'oracle': host logic over the CPU oracle (runs anywhere);
    'hip': the real product path on the MI355X (gpu-marked)."""
    if request.param == "oracle":
        install_oracle_backend(monkeypatch)
    else:
        import torch

        assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    return request.param


@pytest.fixture
def galois_field():
    from honeybadgermpc_amd.field import GF

    return GF(BLS)


@pytest.fixture
def galois_field_roots(golden):
    """2^r-th roots of unity of BLS12-381's scalar field, r = 0..10, from the golden file
    (the reference's fixture list, tests/fixtures.py:23-57, holds 33 of them; the first
    entries are checked against literals in test_oracle_golden.py)."""
    c = golden("constants.json")
    return [1] + [c["omega"][str(1 << r)] for r in range(1, 11)]


@pytest.fixture
def polynomial(galois_field):
    from honeybadgermpc_amd.polynomial import polynomials_over

    return polynomials_over(galois_field)


def set_hook(monkeypatch, name, value):
    """flip an environment hook of the library inside this process: the library reads its hooks once (hb_common.hpp, env_hook), so it is
    told to read them again -- and again when the test is over (the fixture below)"""
    monkeypatch.setenv(name, value)
    _reload_env()


def clear_hook(monkeypatch, name):
    monkeypatch.delenv(name, raising=False)
    _reload_env()


def _reload_env():
    from honeybadgermpc_amd import _capi

    if _capi._lib is not None:
        _capi._lib.hb_debug_reload_env()


@pytest.fixture(autouse=True)
def _hooks_as_the_environment_says():
    yield
    _reload_env()          # (torn down after monkeypatch has restored the environment)
