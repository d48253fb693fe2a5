"""Run representative tile-GEMM shapes of the PaDT_Pro_3B step (for rocprofv3 --pmc / timing).
usage: python tools/bench_gemm_tile.py [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from padt_amd import ops

BF = torch.bfloat16
SHAPES = [  # (M, N, K, epilogue)
    (16928, 3840, 1280, 0), (16928, 1280, 1280, 2), (16928, 6912, 1280, 3), (16928, 1280, 3456, 2),
    (4616, 2560, 2048, 0), (4616, 2048, 2048, 2), (4616, 22016, 2048, 3), (4616, 2048, 11008, 2),
    (8192, 8192, 8192, 0),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    knob = sys.argv[2] if len(sys.argv) > 2 else None             # optional env knob to A/B inside one process: NAME=v0,v1
    if knob:
        name, vals = knob.split("=")
        for (M, N, K, epi) in SHAPES:
            a = torch.randn(M, K, device="cuda").to(BF)
            w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
            out = torch.zeros(M, N // 2 if epi == 3 else N, device="cuda", dtype=BF)
            res = out if epi == 2 else None
            line = f"M={M:6d} N={N:6d} K={K:6d} epi={epi}:"
            for v in vals.split(",") * 2:
                os.environ[name] = v
                for _ in range(2):
                    ops.gemm(a, w, out=out, epilogue=epi, residual=res)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.gemm(a, w, out=out, epilogue=epi, residual=res)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                line += f"  {name}={v}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
            print(line, flush=True)
        return
    for (M, N, K, epi) in SHAPES:
        a = torch.randn(M, K, device="cuda").to(BF)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
        out = torch.zeros(M, N // 2 if epi == 3 else N, device="cuda", dtype=BF)
        res = out if epi == 2 else None
        for _ in range(2):
            ops.gemm(a, w, out=out, epilogue=epi, residual=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm(a, w, out=out, epilogue=epi, residual=res)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"M={M:6d} N={N:6d} K={K:6d} epi={epi}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
