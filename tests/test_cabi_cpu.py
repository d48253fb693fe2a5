"""CPU-side checks of the C-ABI boundary: the in-tree library loads, exports every symbol include/padt_hip.h declares,
and rejects malformed arguments before touching a device (no compute calls without a GPU)."""
import os

import pytest

from padt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from padt_amd.build import build
        build()
    return _lib.load()


def test_header_parses_and_every_symbol_is_exported(lib):
    decls = _lib.parse_header()
    assert len(decls) >= 26
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in include/padt_hip.h but not exported"
    assert lib.padt_abi_version() == 1


def test_every_extern_c_symbol_is_declared(lib):
    """No undeclared entry points: every padt_* extern "C" definition in csrc/ appears in the header."""
    import re
    decls = set(_lib.parse_header())
    defined = set()
    csrc = os.path.join(ROOT, "padt_amd", "csrc")
    for f in os.listdir(csrc):
        src = open(os.path.join(csrc, f)).read()
        defined |= set(re.findall(r'extern "C"\s+[\w\s\*]+?\b(padt_\w+)\s*\(', src))
    defined -= {"padt_set_error", "padt_gemm256_try", "padt_gemm_fp8_impl"}      # internal helpers shared between translation units
    assert defined <= decls, defined - decls


def test_argument_validation_without_device(lib):
    # K not a multiple of 8 → -1 with a message, before any launch
    st = lib.padt_gemm_bf16(None, 16, 100, 16, 100, None, 16, 8, None, 0, 4, 8, 100, 0, 0, None)
    assert st == -1 and b"multiples of 8" in lib.padt_last_error()
    st = lib.padt_attn_varlen(None, 16, 7, 16, 8, 16, 8, 16, 8, None, None, 1, 4, 2, 2, 80, 0.1, 0, None, None, 0)
    assert st == -1
    st = lib.padt_decode_attn(None, 16, 16, 16, None, 16, 16, 1, 16, 2, 128, 100, 50, 0.1)   # s_max % 64 != 0
    assert st == -1
    assert lib.padt_vrt_head_nblk(151936, 529) == (151936 + 529 + 15) // 16
    assert lib.padt_decode_attn_workspace(8, 2, 128, 640) == 8 * 2 * 10 * 16 * 130 * 4
    # zero-sized work is a no-op success
    assert lib.padt_gemm_bf16(None, None, 8, None, 8, None, None, 8, None, 0, 0, 8, 8, 0, 0, None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PaDTHipError, match="no CPU/PyTorch fallback"):
        _lib.load()
