"""Do the decode projections run faster when their weights are already in the 256 MB Infinity Cache (memory-side, in front of HBM)?

The question behind DESIGN.md §6.5 item 5: a decode step is a chain of ~220 dependent launches, most of them too short to pull HBM at full
rate; a side branch of the step's hipGraph could stream the NEXT projections' weights (gate/up + down of a layer = 135 MB) through the chip
while the attention / merge / o launches of the same layer run (≈25 us, little HBM traffic), so that gate/up and down then read from the
Infinity Cache.  This script measures the ceiling of that idea with the product kernels, nothing new built:
  cold   6 rotating weight sets (540 MB: every launch misses the cache)                 — what a decode step sees today
  warm   touch(w_i) [a torch reduction over the packed weight image] immediately before gemm(w_i), same rotation; the touch alone is timed
         too, so   warm gemm = (touch + gemm) - touch
  hot    1 weight set (what a naive micro-benchmark reports)
at 8 and 64 rows, launches replayed from one hipGraph; and then the idea itself on one layer's launch sequence (6 rotating layers of weights,
924 MB): qkv → [rope + append + split attention, merge] → o → gate/up → down on the main stream, with and without a forked branch that
touches that layer's gate/up and down images while attention / merge / o run.

    python tools/bench_mall_warm.py            (GPU box; ≈40 s after import)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from padt_amd import ops  # noqa: E402

DT = torch.float16


def graph_time(fn, n):
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def main():
    D, I, QKV = 2048, 11008, 2560
    n = 120
    print(f"{'rows':>4s} {'proj':5s} {'MB':>6s} | {'cold us':>8s} {'TB/s':>5s} | {'touch us':>8s} {'touch+gemm':>10s} {'warm us':>8s} {'TB/s':>5s} | {'hot us':>7s} {'TB/s':>5s}")
    for B in (8, 64):
        B16 = (B + 15) // 16 * 16
        for name, N, K, epi in (("qkv", QKV, D, 0), ("o", D, D, 2), ("gu", 2 * I, D, 3), ("down", D, I, 2)):
            nset = max(6, int(640e6 / (N * K * 2)) + 1)             # the rotation must exceed the 256 MB cache for the small matrices too
            ws = [ops.pack_weight((torch.randn(N, K, device="cuda") * 0.02).to(DT)) for _ in range(nset)]
            a = torch.randn(B, K, device="cuda").to(DT)
            a_pk = torch.zeros(B16, K, device="cuda", dtype=DT)
            ops.pack_rows(a, a_pk, B, to_packed=True)
            n_out = N // 2 if epi == 3 else N
            out_pk = torch.zeros(B16, n_out, device="cuda", dtype=DT)
            out = torch.zeros(B, n_out, device="cuda", dtype=DT)
            split = 2 if name == "down" else 1
            wsp = ops.new_splitk_workspace(N, 2, "cuda")
            i = [0]

            def gemm(w):
                if epi == 2:
                    ops.gemm_packed(a_pk, w, N, out=out_pk, epilogue=2, residual=out_pk, split_k=split, workspace=wsp, a_packed=True, c_packed=True, rows=B)
                else:
                    ops.gemm_packed(a_pk, w, N, out=out_pk if epi == 3 else out, epilogue=epi, norm_eps=1e-6, a_packed=True, c_packed=(epi == 3), rows=B)

            def touch(w):
                w.view(torch.int32).max()                          # one streaming read of the packed image

            def cold():
                i[0] += 1
                gemm(ws[i[0] % nset])

            def touch_only():
                i[0] += 1
                touch(ws[i[0] % nset])

            def warm():
                i[0] += 1
                touch(ws[i[0] % nset])
                gemm(ws[i[0] % nset])

            def hot():
                gemm(ws[0])

            mb = N * K * 2 / 1e6
            tc, tt, tw, th = graph_time(cold, n), graph_time(touch_only, n), graph_time(warm, n), graph_time(hot, n)
            print(f"{B:4d} {name:5s} {mb:6.1f} | {tc:8.2f} {mb / tc:5.2f} | {tt:8.2f} {tw:10.2f} {tw - tt:8.2f} {mb / max(tw - tt, 1e-3):5.2f} | {th:7.2f} {mb / th:5.2f}", flush=True)
            del ws
    layer_sequence()


def layer_sequence(B=64, n_layers=36):
    D, I, QKV, Hq, Hkv, HD, S_max = 2048, 11008, 2560, 16, 2, 128, 640
    mk = lambda N, K: ops.pack_weight((torch.randn(N, K, device="cuda") * 0.02).to(DT))
    sets = [(mk(QKV, D), mk(D, D), mk(2 * I, D), mk(D, I)) for _ in range(6)]
    x_pk = torch.zeros(B, D, device="cuda", dtype=DT)
    h_pk = torch.zeros(B, I, device="cuda", dtype=DT)
    ops.pack_rows(torch.randn(B, D, device="cuda").to(DT), x_pk, B, to_packed=True)
    qkv = torch.zeros(B, QKV, device="cuda", dtype=DT)
    att = torch.zeros(B, Hq * HD, device="cuda", dtype=DT)
    kcs = [torch.randn(B, Hkv, S_max, HD, device="cuda").to(DT) for _ in range(6)]
    vts = [torch.randn(B, Hkv, HD, S_max, device="cuda").to(DT) for _ in range(6)]
    cs = torch.randn(B, HD // 2, 2, device="cuda")
    slot = torch.full((B,), 600, dtype=torch.int32, device="cuda")
    wsd = ops.new_decode_workspace(B, Hkv, HD, S_max, "cuda")
    wsp = ops.new_splitk_workspace(D, 2, "cuda")
    side = torch.cuda.Stream()

    def step(prefetch):
        main = torch.cuda.current_stream()
        for l in range(n_layers):
            wq, wo, wgu, wd = sets[l % 6]
            ops.gemm_packed(x_pk, wq, QKV, out=qkv, epilogue=0, norm_eps=1e-6, a_packed=True, c_packed=False, rows=B)
            if prefetch:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    wgu.view(torch.int32).max()
                    wd.view(torch.int32).max()
            ops.decode_attn_rope(qkv, cs, slot, kcs[l % 6], vts[l % 6], att, wsd, Hq, Hkv, HD, S_max, S_max)
            ops.gemm_packed(x_pk, wo, D, out=x_pk, epilogue=2, residual=x_pk, a_packed=True, c_packed=True, rows=B)
            if prefetch:
                main.wait_stream(side)
            ops.gemm_packed(x_pk, wgu, 2 * I, out=h_pk, epilogue=3, norm_eps=1e-6, a_packed=True, c_packed=True, rows=B)
            ops.gemm_packed(h_pk, wd, D, out=x_pk, epilogue=2, residual=x_pk, split_k=2, workspace=wsp, a_packed=True, c_packed=True, rows=B)

    res = {}
    for pf in (False, True, False, True):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            t = graph_time(lambda: step(pf), 1) / n_layers
        res.setdefault(pf, []).append(t)
    print(f"layer sequence at {B} rows (qkv, attention + merge, o, gate/up, down), us per layer: plain {res[False]}, with the prefetch branch {res[True]}")


if __name__ == "__main__":
    main()
