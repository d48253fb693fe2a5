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
    twins = _lib.parse_header(_lib.HEADER_F16)
    assert len(decls) >= 26 and len(twins) >= 20
    for name in list(decls) + list(twins):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert lib.padt_abi_version() == 4
    assert lib.padt_stream_scale(0) == 1.0 and lib.padt_stream_scale(1) == 2.0 ** -4


def test_fp16_header_is_what_the_generator_writes():
    """include/padt_hip_f16.h is generated (tools/gen_f16_header.py): same argument lists as the bf16 declarations, one twin per
    PADT_TWIN / PADT_SYM definition in csrc/."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_f16_header", os.path.join(ROOT, "tools", "gen_f16_header.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert open(_lib.HEADER_F16).read() == gen.render()
    bf, f16 = _lib.parse_header(), _lib.parse_header(_lib.HEADER_F16)
    for b, f in gen.twins():
        assert bf[b][:2] == f16[f][:2], (b, f)


def test_every_extern_c_symbol_is_declared(lib):
    """No undeclared entry points: every padt_* extern "C" definition in csrc/ appears in the header."""
    import re
    decls = set(_lib.parse_header()) | set(_lib.parse_header(_lib.HEADER_F16))
    defined = set()
    csrc = os.path.join(ROOT, "padt_amd", "csrc")
    for f in os.listdir(csrc):
        src = open(os.path.join(csrc, f)).read()
        defined |= set(re.findall(r'extern "C"\s+[\w\s\*]+?\b(padt_\w+)\s*\(', src))
        for nm in re.findall(r'extern "C"\s+[\w\s\*]+?\bPADT_TWIN\((padt_\w+)\)\s*\(', src):      # both operand-type instantiations
            defined |= {nm, nm + "_f16"}
        for pre, post in re.findall(r'extern "C"\s+[\w\s\*]+?\bPADT_SYM\((padt_\w+),\s*(\w*)\)\s*\(', src):
            defined |= {pre + "bf16" + post, pre + "f16" + post}
    internal = {"padt_set_error", "padt_gemm256_try", "padt_gemm_fp8_impl"}      # helpers shared between translation units
    defined -= internal | {n + "_f16" for n in internal}
    assert defined <= decls, defined - decls
    # and nothing else leaves the library: every exported padt_* symbol is declared (or one of the internal helpers)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if out.returncode == 0:
        exported = {l.split()[-1] for l in out.stdout.splitlines() if l.split() and l.split()[-1].startswith("padt_") and " T " in l}
        extra = exported - decls - internal - {n + "_f16" for n in internal}
        assert not extra, extra


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


def test_integration_stub_binds_the_declared_signature():
    """INTEGRATION.md shows the ctypes stub a PaDT maintainer adds for flash_attn_varlen_func: its argtypes list and its call must have as
    many arguments as include/padt_hip.h declares for padt_attn_varlen (round 3 shipped a stub three arguments short)."""
    import re
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    n_decl = len(_lib.parse_header()["padt_attn_varlen"][1])
    m = re.search(r"_lib\.padt_attn_varlen\.argtypes = \[([^\]]*)\]", doc)
    assert m and len([a for a in m.group(1).split(",") if a.strip()]) == n_decl
    call = doc[doc.index("st = _lib.padt_attn_varlen("):]
    call = call[: call.index("\n    if st != 0")]
    depth, n_args = 0, 1
    for ch in call[call.index("(") + 1:]:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            if depth == 0:
                break
            depth -= 1
        elif ch == "," and depth == 0:
            n_args += 1
    assert n_args == n_decl, (n_args, n_decl)
