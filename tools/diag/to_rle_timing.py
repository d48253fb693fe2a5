"""Where does the to_rle leg lose time at 20 steps?  Re-runs bench.py's to_rle loop with host timers around the post-processing and
compares: (a) no post-processing, (b) post-processing inline (the leg), (c) post-processing deferred to after the flush."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench  # noqa: E402


def main():
    steps = int(os.environ.get("STEPS", "20"))
    sys.argv = [sys.argv[0], "--steps", str(steps), "--warmup", "5"]
    args = bench.parse_args()
    cfg, model, grid_hw = bench.build_model(args, "cuda:0")
    inp = bench.make_inputs(cfg, args, grid_hw, "cuda:0", seed=1234, dtype=model.dtype)
    from padt_amd import pipeline, postprocess
    sizes = [(640, 640)] * args.batch
    post_stream = torch.cuda.Stream(priority=-1)
    for mode in ("none", "inline", "inline", "deferred", "none"):
        r = pipeline.PipelinedRunner(model, inp["proc"], depth=args.depth, merge=args.merge)
        t_post = [0.0]
        held = []

        def post(done):
            if mode == "none":
                return
            if mode == "deferred":
                held.extend(done)
                return
            t0 = time.perf_counter()
            for decoded, completions, labels, vrts in done:
                with torch.cuda.stream(post_stream):
                    postprocess.postprocess_results(decoded, labels, sizes, want_mask=False)
            t_post[0] += time.perf_counter() - t0

        def go(k):
            for _ in range(k):
                ids, am, pix = bench.next_batch(inp)
                post(r.submit(ids, am, pix, inp["grid"], max_new_tokens=args.tnew, schedule=inp["sched"]))
            post(r.flush())
        go(args.depth * args.merge)
        held.clear()
        t_post[0] = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(steps)
        torch.cuda.synchronize()
        e_run = time.perf_counter() - t0
        if mode == "deferred":
            t1 = time.perf_counter()
            for decoded, completions, labels, vrts in held:
                postprocess.postprocess_results(decoded, labels, sizes, want_mask=False)
            torch.cuda.synchronize()
            t_post[0] = time.perf_counter() - t1
        e = time.perf_counter() - t0
        print(f"steps {steps} mode {mode:9s}: {args.batch * steps / e:7.2f} images/s  ({e * 1e3:7.1f} ms total, runner part {e_run * 1e3:7.1f} ms, host time inside post {t_post[0] * 1e3:6.1f} ms = {t_post[0] / steps * 1e3:.2f} ms per batch)")


if __name__ == "__main__":
    main()
