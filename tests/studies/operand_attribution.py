"""Hand-run study (not collected by pytest): WHICH operand class sets the end-to-end distance to the fp32 oracle at full PaDT_Pro_3B
depth, and what fp16 operands (`v_mfma_f32_16x16x32_f16`: same rate as bf16 on gfx950, 3 more mantissa bits) would give.

The oracle is run teacher-forced on the fp32 run's tokens (inputs of test_full_depth_3b_teacher_forced_against_oracle: one 46 x 46
image, 8 steps, 4 VRT) under parity_util.operand_floor(dtype, classes) for
  * every class alone at bf16 (attribution),
  * all classes at bf16 / at fp16 (the two floors),
  * the norm-folded weight images the HIP path multiplies with (round(w_norm * W) in bf16 / fp16 instead of W), alone and together with
    the matching floor — the one weight-side rounding the HIP path adds,
and every run goes through the fp32 PaDT decoder.  ≈1.3 min per run on 8 cores, 25 GB.

    python tests/studies/operand_attribution.py [out.md]
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import parity_util as U  # noqa: E402

O = U.O


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item(), ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


def folded(w, cfg, dt):
    return U.folded_weight_images(w, cfg, dt)


def main():
    import padt_amd
    from padt_amd.weights import synthetic_state_dict
    cfg = padt_amd.padt_pro_3b()
    sd = synthetic_state_dict(cfg, seed=3, std=0.02, bias_std=0.02, norm_jitter=0.1, device="cpu", dtype=torch.bfloat16)
    w = {k: v.float() for k, v in sd.items()}
    del sd
    oc = U.oracle_config(cfg)
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 46, 46]], n_pre=15, n_post=33, seed=77)
    T = 8
    sched = U.rec_schedule(T, vrt_at=range(2, 6))
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    rows = []

    def decode(r):
        st = r["state"]
        feats = [[torch.cat([r["hidden"][t][0:1, -1] for t in range(2, 6)], 0)]]
        return O.vl_decode(w, oc, feats, st.proto, st.high_res, grid, st.visual_pe)

    with torch.no_grad():
        t0 = time.perf_counter()
        ref = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True)
        toks = ref["sequences"][:, ids.shape[1]:]
        ref_out = decode(ref)
        print(f"fp32 oracle: {time.perf_counter() - t0:.1f} s", flush=True)

        def run(label, dt, classes, weights=None):
            t0 = time.perf_counter()
            with U.operand_floor(dt, classes):
                px = pix.to(dt).float() if "vit.misc" in classes else pix
                r = O.generate(weights or w, oc, ids, am, px, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
            o = decode(r)
            hid = max(rel(r["hidden"][t][:, -1], ref["hidden"][t][:, -1])[1] for t in range(T))
            lg = 0.0
            for t in range(T):
                a, b = r["logits"][t][0], ref["logits"][t][0]
                fin = torch.isfinite(b)
                lg = max(lg, ((a[fin] - b[fin]).abs().max() / b[fin].abs().max()).item())
            row = (label, rel(r["state"].high_res, ref["state"].high_res)[1], hid, lg,
                   (o["pred_boxes"] - ref_out["pred_boxes"]).abs().max().item(),
                   (o["pred_score"] - ref_out["pred_score"]).abs().max().item(), *rel(o["pred_mask"], ref_out["pred_mask"]))
            rows.append(row)
            print("%-28s vit %.2e hid %.2e logit %.2e box %.2e score %.2e mask max %.2e rms %.2e  (%.0f s)" % (*row, time.perf_counter() - t0), flush=True)

        bf, fp = torch.bfloat16, torch.float16
        run("all classes, bf16", bf, U.OPERAND_CLASSES)
        run("all classes, fp16", fp, U.OPERAND_CLASSES)
        for c in U.OPERAND_CLASSES:
            run("bf16: " + c + " alone", bf, (c,))
        run("bf16: all vit.*", bf, [c for c in U.OPERAND_CLASSES if c.startswith("vit.")])
        run("bf16: all llm.* + head", bf, [c for c in U.OPERAND_CLASSES if not c.startswith("vit.")])
        wf = folded(w, cfg, bf)
        run("folded weights bf16 alone", bf, (), wf)
        run("all bf16 + folded bf16", bf, U.OPERAND_CLASSES, wf)
        del wf
        wf = folded(w, cfg, fp)
        run("folded weights fp16 alone", fp, (), wf)
        run("all fp16 + folded fp16", fp, U.OPERAND_CLASSES, wf)
        del wf

    md = ["| run | ViT high_res rel rms | hidden rows rel rms (worst step) | logits abs max / largest logit | boxes abs max | score abs | mask logits max / range | mask rel rms |",
          "|---|---|---|---|---|---|---|---|"]
    md += ["| %s | %.2e | %.2e | %.2e | %.2e | %.2e | %.2e | %.2e |" % r for r in rows]
    text = "\n".join(md)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("# Operand-class attribution at full PaDT_Pro_3B depth (tests/studies/operand_attribution.py; oracle only, CPU)\n\n" + text + "\n")


if __name__ == "__main__":
    main()
