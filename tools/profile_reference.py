"""One batch of the bench workload (8 images, T_new = 28) through precision="reference" for a kernel trace:
    cd /tmp && rocprofv3 --kernel-trace -d out -o trace -- python $REPO/tools/profile_reference.py   (then tools/rocpd_stats.py)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "0"]
    args = bench.parse_args()
    import padt_amd
    from padt_amd import pipeline
    from padt_amd.modeling import PaDTForConditionalGeneration
    cfg = padt_amd.padt_pro_3b()
    model = PaDTForConditionalGeneration.from_synthetic(cfg, seed=0, device="cuda:0", operands="fp16", precision="reference")
    args.operands, args.policy = "fp16", "fp16"
    inp = bench.make_inputs(cfg, args, (46, 46), "cuda:0", seed=1234, dtype=torch.float16)
    for k in range(3):
        ids, am, pix = bench.next_batch(inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipeline.rec_batch(model, inp["proc"], ids, am, pix, inp["grid"], max_new_tokens=args.tnew, schedule=inp["sched"])
        torch.cuda.synchronize()
        print(f"batch {k}: {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
