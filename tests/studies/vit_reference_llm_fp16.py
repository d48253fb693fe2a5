"""Hand-run study (not collected by pytest; oracle only, CPU): what a HYBRID mode would give — the ViT on the reference-precision machinery (exact here),
the LLM on the default path's fp16 MFMA operands with its norm-folded fp16 weight images — at full PaDT_Pro_3B depth.  Same inputs and read-out as
operand_attribution.py.  The ViT carries 3/4 of the default path's mask-logit distance (profiles/r04_operand_attribution.md) and is 40 % of its time.

    python tests/studies/vit_reference_llm_fp16.py [out.md]
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
        ref = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True)
        toks = ref["sequences"][:, ids.shape[1]:]
        ref_out = decode(ref)

        def run(label, dt, classes, weights=None):
            t0 = time.perf_counter()
            with U.operand_floor(dt, classes):
                r = O.generate(weights or w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
            o = decode(r)
            hid = max(rel(r["hidden"][t][:, -1], ref["hidden"][t][:, -1])[1] for t in range(T))
            row = (label, rel(r["state"].high_res, ref["state"].high_res)[1], hid, (o["pred_boxes"] - ref_out["pred_boxes"]).abs().max().item(),
                   (o["pred_score"] - ref_out["pred_score"]).abs().max().item(), *rel(o["pred_mask"], ref_out["pred_mask"]))
            rows.append(row)
            print("%-44s vit %.2e hid %.2e box %.2e score %.2e mask max %.2e rms %.2e  (%.0f s)" % (*row, time.perf_counter() - t0), flush=True)

        fp = torch.float16
        llm = [c for c in U.OPERAND_CLASSES if not c.startswith("vit.")]
        run("LLM classes fp16, weights unfolded", fp, llm)
        wf = U.folded_weight_images(w, cfg, fp)
        wl = {k: (w[k] if k.startswith("visual.") else wf[k]) for k in w}
        del wf
        run("LLM classes fp16 + folded LLM weights fp16", fp, llm, wl)
        run("folded LLM weights fp16 alone", fp, (), wl)
    md = ["| run (3b, one 46 x 46 image, 8 steps; ViT exact) | ViT high_res rel rms | hidden rows rel rms (worst step) | boxes abs max | score abs | mask logits max / range | mask rel rms |",
          "|---|---|---|---|---|---|---|"]
    md += ["| %s | %.2e | %.2e | %.2e | %.2e | %.2e | %.2e |" % r for r in rows]
    text = "\n".join(md)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("# Hybrid floor: reference-precision ViT + fp16 LLM at full PaDT depth (tests/studies/vit_reference_llm_fp16.py; oracle only, CPU)\n\n" + text + "\n")


if __name__ == "__main__":
    main()
