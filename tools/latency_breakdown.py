"""Where does a LONE batch's latency go?  (bench.py single_batch_latency: ≈112 ms against ≈67 ms of kernels.)
Phases of rec_batch for PaDT_Pro_3B, batch 8, REC T=28, each timed twice: host time to ENQUEUE (no sync) and wall time until the
GPU is idle.  Run on the GPU box:  python tools/latency_breakdown.py [--model 3b|small]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import padt_amd  # noqa: E402
from padt_amd import pipeline  # noqa: E402
from padt_amd.modeling import PaDTForConditionalGeneration  # noqa: E402
from padt_amd.processor import parseVRTintoCompletion  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from padt_amd.synthetic import FakeProcessor, rec_schedule, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="3b")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    cfg = {"3b": padt_amd.padt_pro_3b, "small": padt_amd.small_test_config}[a.model]()
    g = (46, 46) if a.model == "3b" else (10, 12)
    model = PaDTForConditionalGeneration.from_synthetic(cfg, seed=0, device="cuda:0")
    B, T = 8, 28
    grid, pix, ids, am = synthetic_batch(cfg, [[1, g[0], g[1]]] * B, n_pre=15, n_post=33, seed=1)
    pix, ids, am = pix.cuda().to(torch.bfloat16), ids.cuda(), am.cuda()
    proc = padt_amd.VisonTextProcessingClass(FakeProcessor(cfg, g[0] * g[1] // 4), 2)
    proc.model_embed_token_size = cfg.vocab_size
    sched = rec_schedule(T, range(11, 16))
    sync = torch.cuda.synchronize
    for _ in range(3):
        pipeline.rec_batch(model, proc, ids.clone(), am, pix, grid, max_new_tokens=T, schedule=sched, sync_every=T)
    sync()
    acc = {}

    def phase(name, fn):
        sync()
        t0 = time.perf_counter()
        r = fn()
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        e = acc.setdefault(name, [0.0, 0.0])
        e[0] += (t1 - t0) * 1e3
        e[1] += (t2 - t0) * 1e3
        return r

    for _ in range(a.reps):
        L = ids.shape[1]
        gids = phase("assign_to_global", lambda: proc.assign_to_global_vrt_id(ids.clone(), grid))
        ctx = phase("generate_launch (plan + ViT + prefill + decode chunk)", lambda: model.generate_launch(gids, am, pix, grid, T, False, sched, T, True, 0))
        out = phase("generate_collect", lambda: model.generate_collect(ctx))
        seq = phase("sequences.cpu + assign_to_local", lambda: proc.assign_to_local_vrt_id(out["sequences"].cpu(), grid.cpu()))
        parsed = phase("parseVRTintoCompletion", lambda: parseVRTintoCompletion(proc, seq[:, L:], out["hidden_states"], torch.Tensor([False] * B)))
        phase("vl_decode", lambda: model.vl_decode(parsed[1], out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe))
        # finer: the pieces of generate_launch, run stand-alone
        from padt_amd.llm import plan_prompt
        phase("  plan_prompt (host ints + H2D)", lambda: plan_prompt(cfg, gids, am, grid, model.device))
        phase("  ViT", lambda: model.visual(pix, grid))
    tot_h = tot_w = 0.0
    for k, (h, w) in acc.items():
        print(f"{k:60s} host {h / a.reps:8.2f} ms   until idle {w / a.reps:8.2f} ms")
        if not k.startswith("  "):
            tot_h += h / a.reps
            tot_w += w / a.reps
    print(f"{'sum of the top-level phases':60s} host {tot_h:8.2f} ms   until idle {tot_w:8.2f} ms")
    sync()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        pipeline.rec_batch(model, proc, ids.clone(), am, pix, grid, max_new_tokens=T, schedule=sched, sync_every=T)
    sync()
    print(f"rec_batch back to back: {(time.perf_counter() - t0) / a.reps * 1e3:.2f} ms per batch")


if __name__ == "__main__":
    main()
