"""Build the gfx950 C-ABI shared library in-tree:  python -m padt_amd.build

Plain hipcc (no torch headers): the library's boundary is ``include/padt_hip.h`` — raw device pointers and a
hipStream_t in, status out.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpadt_hip.so")
SOURCES = ["capi.hip", "gemm.hip", "gemm256.hip", "attention.hip", "elementwise.hip", "vrt_head.hip", "decoder_hp.hip", "resize.hip", "rle.hip"]
# Translation units written against the 16-bit operand type X of csrc/common.h: compiled a second time with X = fp16 (-DPADT_OP16_F16=1;
# own namespace, entry points suffixed _f16: include/padt_hip_f16.h) and linked into the same library.
TWIN_SOURCES = ["gemm.hip", "gemm256.hip", "attention.hip", "elementwise.hip", "vrt_head.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# Per-source extras (dropped with a warning if this hipcc does not know them).  -amdgpu-mfma-vgpr-form: MFMA results land in ordinary
# VGPRs instead of AccVGPRs.  hipcc's default keeps the attention kernels' score / output tiles in AccVGPRs and moves every value to a
# VGPR and back for the softmax (184 v_accvgpr_read/write per 44 MFMAs in the ViT full-attention loop) — with the LDS-DMA K / V staging
# the full-attention kernel drops from 212 to 160 registers (three waves per SIMD instead of two): 373 → 323 us; the prompt kernel
# (LDS-bound at two blocks per CU) loses 3 %, the decode kernels are unchanged (profiles/r02_pmc_attention.md).
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(job):
        src, f16 = job
        obj = os.path.join(objdir, src.replace(".hip", "_f16.o" if f16 else ".o"))
        defs = ["-DPADT_OP16_F16=1"] if f16 else []
        cmd = [hipcc, *FLAGS, *defs, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 and src in EXTRA_FLAGS:
            print(f"[padt_amd.build] {src}: extra flags {EXTRA_FLAGS[src]} rejected, compiling without them", file=sys.stderr)
            cmd = [hipcc, *FLAGS, *defs, "-c", os.path.join(CSRC, src), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    jobs = [(s, False) for s in SOURCES] + [(s, True) for s in TWIN_SOURCES]
    jobs.sort(key=lambda j: j[0] not in ("gemm256.hip", "gemm.hip", "attention.hip"))      # longest compiles first
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(jobs))) as ex:
        objs = list(ex.map(compile_one, jobs))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
