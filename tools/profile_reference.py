"""The bench workload (batches of 8 images, T_new = 28) through precision="reference" on the bench leg's runner (32 batches) for a kernel trace:
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
    # the bench leg's runner: two decode groups of `merge` batches in flight, captured decode steps
    r = pipeline.PipelinedRunner(model, inp["proc"], depth=args.depth, merge=args.merge)

    def go(n):
        for _ in range(n):
            ids, am, pix = bench.next_batch(inp)
            r.submit(ids, am, pix, inp["grid"], max_new_tokens=args.tnew, schedule=inp["sched"])
        r.flush()
    go(args.depth * args.merge)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2 * args.merge
    go(n)
    torch.cuda.synchronize()
    e = time.perf_counter() - t0
    print(f"{n} batches: {e / n * 1e3:.1f} ms per batch = {args.batch * n / e:.1f} images/s", file=sys.stderr)


if __name__ == "__main__":
    main()
