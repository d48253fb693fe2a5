"""Hand-run study (not collected by pytest): the fp16-operand floor (parity_util.operand_floor + folded fp16 weight images) at the OVD geometry of
test_3b_ovd_geometry_merged_runner_against_oracle — PaDT_Pro_3B full depth, ONE 46 x 46 image, the 80-class prompt (L = 890), 120 teacher-forced
steps, 7 objects x 5 VRT — against the fp32 oracle: what the mask-logit bound of that test (8e-3 of the logit range) is 1.5 x of.
≈5 min on 8 cores, 30 GB.      python tests/studies/ovd_length_floor.py [out.md]
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import parity_util as U  # noqa: E402
from padt_amd.synthetic import multi_object_schedule  # noqa: E402

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
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 46, 46]], n_pre=15, n_post=346, seed=700)
    T, n_obj, n_vrt = 120, 7, 5
    sched = multi_object_schedule(T, n_obj=n_obj, n_vrt=n_vrt)
    runs = [[t for t in range(T) if sched[t] == "v"][k * n_vrt: (k + 1) * n_vrt] for k in range(n_obj)]
    torch.set_num_threads(min(64, os.cpu_count() or 8))

    def decode(r):
        st = r["state"]
        feats = [[torch.cat([r["hidden"][t][0:1, -1] for t in rr], 0) for rr in runs]]
        return O.vl_decode(w, oc, feats, st.proto, st.high_res, grid, st.visual_pe)

    with torch.no_grad():
        t0 = time.perf_counter()
        ref = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True)
        toks = ref["sequences"][:, ids.shape[1]:]
        ro = decode(ref)
        print(f"fp32 oracle: {time.perf_counter() - t0:.1f} s", flush=True)
        wf = U.folded_weight_images(w, cfg, torch.float16)
        with U.operand_floor(torch.float16):
            flo = O.generate(wf, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
        fo = decode(flo)
    lg = 0.0
    for t in range(T):
        a, b = flo["logits"][t][0], ref["logits"][t][0]
        fin = torch.isfinite(b)
        lg = max(lg, ((a[fin] - b[fin]).abs().max() / b[fin].abs().max()).item())
    hid = max(rel(flo["hidden"][t][:, -1], ref["hidden"][t][:, -1])[1] for t in range(T))
    db = (fo["pred_boxes"] - ro["pred_boxes"]).abs().max().item()
    mx, rms = rel(fo["pred_mask"], ro["pred_mask"])
    text = ("# fp16-operand floor at the OVD geometry (tests/studies/ovd_length_floor.py; oracle only, CPU)\n\n"
            "PaDT_Pro_3B full depth, one 46 x 46 image, L = %d, T = %d teacher-forced steps, %d objects x %d VRT; fp16 activation operands + folded fp16 weight images vs fp32:\n\n"
            "| hidden rows rel rms (worst step) | logits abs max / largest logit (worst step) | %d boxes abs max | mask logits max / range | mask rel rms |\n|---|---|---|---|---|\n"
            "| %.2e | %.2e | %.2e | %.2e | %.2e |\n" % (ids.shape[1], T, n_obj, n_vrt, n_obj, hid, lg, db, mx, rms))
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)


if __name__ == "__main__":
    main()
