"""Micro-benchmark of the decode-step kernels in isolation (back-to-back launches, HIP events).
usage: python tools/bench_kernels.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from padt_amd import ops

BF = torch.bfloat16


def timeit(fn, n=200):
    if os.environ.get('GRAPH'):
        return timeit_graph(fn, n)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def timeit_graph(fn, n):
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(os.environ.get("B", 8))
    g7 = os.environ.get("GEOM") == "7b"                            # PaDT_Pro_7B projections (default: 3B); FP8=1: the fp8 weight images
    fp8 = bool(os.environ.get("FP8"))
    D, I, QKV = (3584, 18944, 4608) if g7 else (2048, 11008, 2560)
    x = torch.randn(B, D, device="cuda").to(BF)
    h = torch.randn(B, I, device="cuda").to(BF)
    res = {}
    for name, N, K, epi, a in (() if os.environ.get("GEMMS_SKIP") else (("qkv", QKV, D, 0, x), ("o", D, D, 2, x), ("gu", 2 * I, D, 3, x), ("down", D, I, 2, h))):
        ws = [torch.randn(N, K, device="cuda").to(BF) * 0.02 for _ in range(int(os.environ.get('ROT', 6)))]   # rotate weights: defeat L2/MALL reuse
        if fp8:                                                   # timing only: random bytes stand in for the e4m3 image
            ws = [torch.randint(0, 120, (N, (K + 63) // 64 * 64), device="cuda", dtype=torch.uint8) for _ in ws]
            sc = torch.ones(N, device="cuda")
        out = torch.zeros(B, N // 2 if epi == 3 else N, device="cuda", dtype=BF)
        i = [0]
        B16 = (B + 15) // 16 * 16
        a_pk = torch.zeros(B16, K, device="cuda", dtype=BF)
        ops.pack_rows(a, a_pk, B, to_packed=True)
        out_pk = torch.zeros(B16, N // 2 if epi == 3 else N, device="cuda", dtype=BF)
        split = int(os.environ.get("SPLIT_" + name.upper(), 1))
        ws_split = ops.new_splitk_workspace(N, max(split, 2), "cuda")

        def f():
            w = ws[i[0] % len(ws)]
            i[0] += 1
            if os.environ.get("MFMA"):
                if epi == 2:
                    ops.gemm(a, w, out=out, epilogue=2, residual=out)
                else:
                    ops.gemm_rmsnorm(a, w, out=out, epilogue=epi)
            elif fp8:
                if epi == 2:
                    ops.gemm_packed_fp8(a_pk, w, sc, N, out=out_pk, epilogue=2, residual=out_pk, split_k=split, workspace=ws_split,
                                        a_packed=True, c_packed=True, rows=B)
                else:
                    ops.gemm_packed_fp8(a_pk, w, sc, N, out=out_pk if epi == 3 else out, epilogue=epi, norm_eps=1e-6, a_packed=True,
                                        c_packed=(epi == 3), rows=B)
            elif os.environ.get("PACK"):                          # fragment-packed activations, as the decode step runs them
                if epi == 2:
                    ops.gemm_packed(a_pk, w, N, out=out_pk, epilogue=2, residual=out_pk, split_k=split, workspace=ws_split,
                                    a_packed=True, c_packed=True, rows=B)
                else:
                    ops.gemm_packed(a_pk, w, N, out=out_pk if epi == 3 else out, epilogue=epi, norm_eps=1e-6, a_packed=True,
                                    c_packed=(epi == 3), rows=B)
            elif epi == 2:
                ops.gemm_packed(a, w, N, out=out, epilogue=2, residual=out, split_k=split, workspace=ws_split)
            else:
                ops.gemm_packed(a, w, N, out=out, epilogue=epi, norm_eps=1e-6)
        t = timeit(f)
        mb = N * K * (1 if fp8 else 2) / 1e6
        print(f"{name:5s} N={N:6d} K={K:6d}: {t:7.2f} us  {mb / t * 1e-3 * 1e3:8.1f} GB/s ({mb:.1f} MB)")
    if os.environ.get("GEMMS_ONLY"):
        return
    # decode attention (rope + append + split attention + merge), Pro_3B heads, ~600 cached tokens
    Hq, Hkv, D, S_max = 16, 2, 128, 640
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device="cuda").to(BF)
    nkv = int(os.environ.get("ROTKV", 8))                         # rotating cache sets: > the 256 MB Infinity Cache at 64 rows
    kvp = bool(os.environ.get("KVP"))                              # KVP=1: fragment-packed caches → the one-launch kernel (what the decode step runs)
    kcs = [torch.randn(B, Hkv, S_max, D, device="cuda").to(BF) for _ in range(nkv)]
    vts = [torch.randn(B, Hkv, D, S_max, device="cuda").to(BF) for _ in range(nkv)]
    if kvp:
        kcs, vts = [ops.pack_k_cache(k) for k in kcs], [ops.pack_vt_cache(v) for v in vts]
    cs = torch.randn(B, D // 2, 2, device="cuda")
    slot = torch.full((B,), 600, dtype=torch.int32, device="cuda")
    att = torch.zeros(B, Hq * D, device="cuda", dtype=BF)
    wsd = ops.new_decode_workspace(B, Hkv, D, S_max, "cuda")
    j = [0]

    def fa():
        j[0] += 1
        ops.decode_attn_rope(qkv, cs, slot, kcs[j[0] % nkv], vts[j[0] % nkv], att, wsd, Hq, Hkv, D, S_max, S_max, cache_packed=kvp)
    print(f"decode_attn_rope B={B}{' packed caches' if kvp else ''}: {timeit(fa):7.2f} us")
    if os.environ.get("ATTN_ONLY"):
        return
    hid = torch.randn(B, 2048, device="cuda").to(BF)
    table = (torch.randn(151936, 2048, device="cuda") * 0.02).to(BF)
    proto = (torch.randn(4232 * max(1, B // 8), 2048, device="cuda") * 0.02).to(BF)
    voff = torch.arange(0, B + 1, dtype=torch.int32, device="cuda") * 529
    nblk = ops.vrt_head_nblk(151936, proto.shape[0])
    pv = torch.zeros(nblk * B, device="cuda")
    pi = torch.zeros(nblk * B, dtype=torch.int32, device="cuda")
    t = timeit(lambda: ops.vrt_head(hid, table, proto, voff, pv, pi, 151645), 50)
    print(f"vrt_head B={B}: {t:7.2f} us  {(table.numel() + proto.numel()) * 2 / t * 1e-3:7.1f} GB/s")
    tp = ops.pack_weight(table)
    hp = torch.zeros((B + 15) // 16 * 16, 2048, device="cuda", dtype=BF)
    ops.pack_rows(hid, hp, B, to_packed=True)
    t = timeit(lambda: ops.vrt_head(hp, table, proto, voff, pv, pi, 151645, table_packed=tp, rows=B), 50)
    print(f"vrt_head packed B={B}: {t:7.2f} us  {(table.numel() + proto.numel()) * 2 / t * 1e-3:7.1f} GB/s")
    # pure streaming read reference: torch sum over a 45 MB bf16 tensor
    big = [torch.randn(2048, 11008, device="cuda").to(BF) for _ in range(6)]
    i = [0]

    def g():
        i[0] += 1
        return big[i[0] % 6].sum()
    t = timeit(g)
    print(f"torch.sum 45MB: {t:.2f} us {45.09 / t * 1e3:.1f} GB/s")


if __name__ == "__main__":
    main()
