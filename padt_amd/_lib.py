"""ctypes binding of libpadt_hip.so, generated from ``include/padt_hip.h`` and its generated fp16 twin ``include/padt_hip_f16.h``
(single source of truth for the C ABI).

The product path has NO fallback: if the shared library is missing or does not load, importing the ops fails loudly
with instructions to build it (``python -m padt_amd.build``).
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "padt_hip.h")
HEADER_F16 = os.path.join(os.path.dirname(HERE), "include", "padt_hip_f16.h")
LIB_PATH = os.environ.get("PADT_HIP_LIB") or os.path.join(HERE, "libpadt_hip.so")

_CTYPE = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float}


class PaDTHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """→ {name: (restype, [argtypes], [argnames])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    src = src.replace('extern "C" {', "").replace("}", "")
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(padt_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _CTYPE[ret.split()[-1]]
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                nm = re.findall(r"(\w+)\s*$", a)[0]
                ty = a[: a.rfind(nm)].strip()
                argnames.append(nm)
                if "*" in ty:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPE[ty.replace("const", "").strip()])
        out[name] = (restype, argtypes, argnames)
    return out


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PaDTHipError(
            f"{LIB_PATH} not found. The PaDT MI355X path has no CPU/PyTorch fallback: build the HIP extension with "
            "`python -m padt_amd.build` (needs hipcc, gfx950).")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise PaDTHipError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes, _) in {**parse_header(), **parse_header(HEADER_F16)}.items():
        fn = getattr(lib, name)           # AttributeError if the .so lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().padt_last_error()
        raise PaDTHipError(f"{what} failed (status {status}): {msg.decode() if msg else ''}")
