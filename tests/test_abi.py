"""CPU: the C-ABI library loads here (no GPU), exports every symbol include/g2048.h declares, the
Python binding covers exactly that set, and the product fails loudly without a device."""
import ctypes as C
import os
import re

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "g2048.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(g2048_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    ge.build_hip()
    from gym2048_amd import _lib
    return _lib.load()


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/g2048.h but not exported"


def test_binding_covers_header():
    from gym2048_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_abi_version(lib):
    from gym2048_amd import _lib
    assert lib.g2048_abi_version() == _lib.ABI_VERSION


def test_struct_layouts():
    from gym2048_amd import _lib
    assert C.sizeof(_lib.StepIO) == 80      # 10 x 8 bytes (the two int32 dtype codes padded)
    assert C.sizeof(_lib.HostIO) == 64
    assert C.sizeof(_lib.Stats) == 176     # 40 + uint32 highest_hist[32] + int64 return_sum


def test_no_gpu_fails_loudly(lib):
    """Without a device g2048_create must return an error (never a silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.g2048_create(16, 0, 1, 0, C.byref(h))
    assert rc < 0 and not h
    assert b"no CPU path" in lib.g2048_last_error()
    from gym2048_amd import G2048Error, Batched2048, Game2048Env
    with pytest.raises(G2048Error):
        Batched2048(16)
    with pytest.raises(G2048Error):
        Game2048Env()


def test_argument_errors(lib):
    assert lib.g2048_create(0, 0, 1, 0, C.byref(C.c_void_p())) == -1          # n = 0
    assert lib.g2048_create(1 << 33, 0, 1, 0, C.byref(C.c_void_p())) == -1    # too many boards
    assert lib.g2048_step(None, None, 1, None) == -1
    assert b"NULL" in lib.g2048_last_error()
    assert lib.g2048_destroy(None) == 0


def test_product_never_imports_oracle():
    """The product package must not reference the oracle (no CPU fallback)."""
    pkg = os.path.join(ROOT, "gym-2048_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "g2048_oracle" not in src or f == "g2048_device.h", f
