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


def stepwise(decoded, sizes):
    """the statements of postprocess_results one by one, each followed by a stream synchronize, printed when slow"""
    from padt_amd import ops
    marks = []
    def lap(name, t0):
        torch.cuda.current_stream().synchronize()
        marks.append((name, round((time.perf_counter() - t0) * 1e3, 2)))
    t = time.perf_counter(); torch.cuda.current_stream().synchronize(); lap("stream sync (nothing enqueued yet)", t)
    masks = decoded["pred_mask"]; dev = masks.device
    t = time.perf_counter(); hs = (decoded["pred_mask_valid_hw"][0].to(torch.int32) * 4).to(dev); ws = (decoded["pred_mask_valid_hw"][1].to(torch.int32) * 4).to(dev); lap("hs / ws", t)
    t = time.perf_counter(); dh = torch.tensor([s[1] for s in sizes], dtype=torch.int32, device=dev); dw = torch.tensor([s[0] for s in sizes], dtype=torch.int32, device=dev); lap("dh / dw (H2D)", t)
    t = time.perf_counter(); b = ops.mask_upsample_binarize(masks.float().contiguous(), hs, ws, dh, dw, 640, 640); lap("upsample", t)
    t = time.perf_counter(); h = ops.mask_rle_launch(b, dh, dw); lap("rle launch", t)
    t = time.perf_counter(); decoded["pred_boxes"].float().cpu(); lap("boxes.cpu", t)
    t = time.perf_counter(); ops.mask_rle_fetch(h); lap("rle fetch", t)
    if sum(m for _, m in marks) > 5:
        print("      slow post call:", marks, flush=True)


def main():
    steps = int(os.environ.get("STEPS", "20"))
    sys.argv = [sys.argv[0], "--steps", str(steps), "--warmup", "5"]
    args = bench.parse_args()
    cfg, model, grid_hw = bench.build_model(args, "cuda:0")
    inp = bench.make_inputs(cfg, args, grid_hw, "cuda:0", seed=1234, dtype=model.dtype)
    from padt_amd import pipeline, postprocess
    sizes = [(640, 640)] * args.batch
    post_stream = torch.cuda.Stream(priority=-1)
    if os.environ.get("NOGC") == "1":
        import gc
        gc.disable()
    if os.environ.get("PHASES") == "1":                                # which phase of a slow post-processing call waits?
        from padt_amd import ops
        def wrap(name):
            f = getattr(ops, name)
            def g(*a, **k):
                t = time.perf_counter()
                r = f(*a, **k)
                dt = (time.perf_counter() - t) * 1e3
                if dt > 3:
                    print(f"      ops.{name}: {dt:.1f} ms", flush=True)
                return r
            setattr(ops, name, g)
        for nm in ("mask_upsample_binarize", "mask_rle_launch", "mask_rle_fetch"):
            wrap(nm)
        import torch as _t
        _cpu = _t.Tensor.cpu
        def cpu(self, *a, **k):
            t = time.perf_counter()
            r = _cpu(self, *a, **k)
            dt = (time.perf_counter() - t) * 1e3
            if dt > 3:
                print(f"      Tensor.cpu {tuple(self.shape)}: {dt:.1f} ms", flush=True)
            return r
        _t.Tensor.cpu = cpu
    for mode in ("none", "inline", "inline", "deferred", "none"):
        r = pipeline.PipelinedRunner(model, inp["proc"], depth=args.depth, merge=args.merge)
        t_post = [0.0]
        held = []
        per_call = []

        def post(done):
            if mode == "none":
                return
            if mode == "deferred":
                held.extend(done)
                return
            t0 = time.perf_counter()
            for decoded, completions, labels, vrts in done:
                t1 = time.perf_counter()
                with torch.cuda.stream(post_stream):
                    if os.environ.get("STEPWISE") == "1":
                        stepwise(decoded, sizes)
                    postprocess.postprocess_results(decoded, labels, sizes, want_mask=False)
                per_call.append(round((time.perf_counter() - t1) * 1e3, 2))
            t_post[0] += time.perf_counter() - t0

        def go(k):
            for _ in range(k):
                ids, am, pix = bench.next_batch(inp)
                post(r.submit(ids, am, pix, inp["grid"], max_new_tokens=args.tnew, schedule=inp["sched"]))
            post(r.flush())
        go(args.depth * args.merge)
        held.clear()
        print('   priming post calls (ms):', per_call)
        per_call.clear()
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
        print("   timed post calls (ms):", per_call)
        print(f"steps {steps} mode {mode:9s}: {args.batch * steps / e:7.2f} images/s  ({e * 1e3:7.1f} ms total, runner part {e_run * 1e3:7.1f} ms, host time inside post {t_post[0] * 1e3:6.1f} ms = {t_post[0] / steps * 1e3:.2f} ms per batch)")


if __name__ == "__main__":
    main()
