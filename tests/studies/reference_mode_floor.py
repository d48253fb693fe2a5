"""Hand-run study (not collected by pytest): what must a "reference precision" mode split?  Same inputs as operand_attribution.py (full-depth
PaDT_Pro_3B oracle, one 46 x 46 image, 8 steps); every GEMM A operand exact (= 16-bit-mantissa (hi, lo) pairs, 7.6e-6) and only the attention
internals rounded to fp16 — the floor of "split-precision GEMMs + the fp16 MFMA attention kernels as they are".

    python tests/studies/reference_mode_floor.py [out.md]
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
        t0 = time.perf_counter()
        ref = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True)
        toks = ref["sequences"][:, ids.shape[1]:]
        ref_out = decode(ref)
        print(f"fp32 oracle: {time.perf_counter() - t0:.1f} s", flush=True)

        def run(label, dt, classes):
            t0 = time.perf_counter()
            with U.operand_floor(dt, classes):
                r = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
            o = decode(r)
            hid = max(rel(r["hidden"][t][:, -1], ref["hidden"][t][:, -1])[1] for t in range(T))
            row = (label, rel(r["state"].high_res, ref["state"].high_res)[1], hid,
                   (o["pred_boxes"] - ref_out["pred_boxes"]).abs().max().item(),
                   (o["pred_score"] - ref_out["pred_score"]).abs().max().item(), *rel(o["pred_mask"], ref_out["pred_mask"]))
            rows.append(row)
            print("%-44s vit %.2e hid %.2e box %.2e score %.2e mask max %.2e rms %.2e  (%.0f s)" % (*row, time.perf_counter() - t0), flush=True)

        fp = torch.float16
        run("fp16: qkv + p (ViT and LLM)", fp, ("vit.qkv", "vit.p", "llm.qkv", "llm.p"))
        run("fp16: qkv + p + ao (ViT and LLM)", fp, ("vit.qkv", "vit.p", "vit.ao", "llm.qkv", "llm.p", "llm.ao"))
        run("fp16: llm qkv + p + ao only (ViT exact)", fp, ("llm.qkv", "llm.p", "llm.ao"))
        run("fp16: vit qkv + p + ao only (LLM exact)", fp, ("vit.qkv", "vit.p", "vit.ao"))

    md = ["| run | ViT high_res rel rms | hidden rows rel rms (worst step) | boxes abs max | score abs | mask logits max / range | mask rel rms |", "|---|---|---|---|---|---|---|"]
    md += ["| %s | %.2e | %.2e | %.2e | %.2e | %.2e | %.2e |" % r for r in rows]
    text = "\n".join(md)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("# What a reference-precision mode must split (tests/studies/reference_mode_floor.py; oracle only, CPU)\n\n" + text + "\n")


if __name__ == "__main__":
    main()
