"""In-situ rate of the tile-GEMM family from a rocprofv3 kernel-trace summary (tools/rocpd_stats.py table) of bench.py:
Σ algorithmic FLOP of the traced batches ÷ Σ duration of every gemm_tile256_kernel / gemm_tile_kernel launch in the trace.
usage: python tools/insitu.py <kernel_stats.md> <alg_tflop_per_step> <launches_per_step> [out.json]"""
import json
import sys


def main(md, tflop_per_step, launches_per_step, out=None):
    calls, total_ms = 0, 0.0
    for line in open(md):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) >= 3 and cells[0].startswith("gemm_tile"):
            calls += int(cells[1])
            total_ms += float(cells[2])
    steps = calls / launches_per_step
    res = {"source": md, "tile_gemm_launches": calls, "batches_in_trace": round(steps, 3), "tile_gemm_total_ms": round(total_ms, 3),
           "avg_launch_us": round(total_ms * 1e3 / max(calls, 1), 2),
           "in_situ_tflops": round(tflop_per_step * steps / (total_ms * 1e-3), 1) if total_ms else None,
           "alg_tflop_per_step": tflop_per_step, "launches_per_step": launches_per_step}
    print(json.dumps(res))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
