"""Device RLE (padt_mask_rle) alone: time per launch for 8 masks of 640 x 640 — noise-like (what random weights produce: ~50 000 runs),
blob-like (a real mask: ~1 000 runs), all-zero — and the host statement for comparison.  gpurun -- python tools/bench_rle.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from padt_amd import ops, postprocess as P  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:640, 0:640]
    base = rng.standard_normal((160, 160))
    noise = np.kron(base, np.ones((4, 4))) + 0.3 * rng.standard_normal((640, 640))           # up-sampled noise: runs of a few pixels
    cases = {"noise (random-weight masks)": (noise > 0).astype(np.uint8),
             "blob (a real mask)": (((yy - 300) ** 2 + (xx - 320) ** 2) < 200 ** 2).astype(np.uint8),
             "empty": np.zeros((640, 640), np.uint8)}
    dh = torch.full((8,), 640, dtype=torch.int32, device="cuda")
    for name, m in cases.items():
        buf = torch.from_numpy(np.stack([m] * 8)).cuda()
        strs = ops.mask_rle(buf, dh, dh)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(n):
            ops.mask_rle(buf, dh, dh)
        ev1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        t1 = time.perf_counter()
        ref = P.rle_string(P.rle_counts(m))
        host = (time.perf_counter() - t1) * 1e3
        assert strs[0] == ref
        print(f"{name:30s} runs {len(P.rle_counts(m)):6d}  string {len(ref):6d} B  device (8 masks, incl. 2 D2H syncs) {wall:7.3f} ms wall, {ev0.elapsed_time(ev1) / n:7.3f} ms GPU;  host statement {host:6.2f} ms per mask")


if __name__ == "__main__":
    main()
