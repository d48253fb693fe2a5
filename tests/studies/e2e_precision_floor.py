"""Hand-run study (not collected by pytest): the END-TO-END distance to the fp32 oracle that bf16 MFMA operands alone impose at full
PaDT_Pro_3B depth (32 ViT blocks + 36 LLM layers + head), on the inputs of test_full_depth_3b_teacher_forced_against_oracle:
the oracle is run twice — plain fp32, and inside parity_util.bf16_operand_floor() — teacher-forced on the fp32 run's tokens, both
through the fp32 PaDT decoder.  ≈3 min on 8 cores, 20 GB.

    python tests/studies/e2e_precision_floor.py
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
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True)
        toks = ref["sequences"][:, ids.shape[1]:]
        print(f"fp32 oracle: {time.perf_counter() - t0:.1f} s", flush=True)
        with U.bf16_operand_floor():
            flo = O.generate(w, oc, ids, am, pix.to(torch.bfloat16).float(), grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
        outs = []
        for r in (ref, flo):
            st = r["state"]
            feats = [[torch.cat([r["hidden"][t][0:1, -1] for t in range(2, 6)], 0)]]
            outs.append(O.vl_decode(w, oc, feats, st.proto, st.high_res, grid, st.visual_pe))
    print("bf16-operand floor vs fp32 oracle (rel max = |d|max / |ref|max, rel rms):")
    print("  ViT high_res  rel max %.3e rms %.3e" % rel(flo["state"].high_res, ref["state"].high_res))
    print("  prototypes    rel max %.3e rms %.3e" % rel(flo["state"].proto, ref["state"].proto))
    for t in range(T):
        lg_r, lg_f = ref["logits"][t][0], flo["logits"][t][0]
        fin = torch.isfinite(lg_r)
        print("  step %d: hidden rel max %.3e rms %.3e | logits |d|max %.3e (|logit|max %.3e)" % (
            t, *rel(flo["hidden"][t][:, -1], ref["hidden"][t][:, -1]), (lg_f[fin] - lg_r[fin]).abs().max().item(), lg_r[fin].abs().max().item()))
    a, b = outs[1], outs[0]
    print("  boxes |d|max %.3e  score |d|max %.3e  mask rel max %.3e rms %.3e" % (
        (a["pred_boxes"] - b["pred_boxes"]).abs().max().item(), (a["pred_score"] - b["pred_score"]).abs().max().item(), *rel(a["pred_mask"], b["pred_mask"])))


if __name__ == "__main__":
    main()
