"""Parity + timing of the PROTOTYPE decode attention (tools/ubench/decode_attn_v2.hip: no LDS, no barrier, transposed scores, probabilities
fed to the second MFMA from registers) against the library's decode_attn_rope on fp16 operands.  Not part of the product; DESIGN.md §6.5 item 1.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPADT_OP16_F16=1 -mllvm -amdgpu-mfma-vgpr-form=1 -shared -fPIC tools/ubench/decode_attn_v2.hip \
          -o tools/ubench/libdecode_attn_v2.so            (cross-compiles without a GPU; the .so travels with gpurun)
    python tools/bench_decode_attn_v2.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from padt_amd import ops  # noqa: E402

H = torch.float16
_vp, _l, _i, _f = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
LIB = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libdecode_attn_v2.so"))
LIB.decode_attn_rope_v2.argtypes = [_vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i]
LIB.decode_attn_rope_v2.restype = _i


LIB.decode_attn_rope_v3.argtypes = [_vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i]
LIB.decode_attn_rope_v3.restype = _i


def attn_v3(qkv, cs, slot, kc, vt, out, Hq, Hkv, D, S_max, packed=False, nw=8, out_packed=False):
    """round 6: ONE launch, nw waves per (kv head, sample), online merge in registers + LDS, 16-bit output rows written directly."""
    st = LIB.decode_attn_rope_v3(torch.cuda.current_stream().cuda_stream, qkv.data_ptr(), qkv.stride(0), cs.data_ptr(), slot.data_ptr(),
                                 kc.data_ptr(), vt.data_ptr(), out.data_ptr(), qkv.shape[0], Hq, Hkv, D, S_max, float(D ** -0.5),
                                 1 if packed else 0, int(nw), 1 if out_packed else 0)
    assert st == 0, st


def v2(qkv, cs, slot, kc, vt, out, ws, Hq, Hkv, D, S_max, max_len, packed=False):
    st = LIB.decode_attn_rope_v2(torch.cuda.current_stream().cuda_stream, qkv.data_ptr(), qkv.stride(0), cs.data_ptr(), slot.data_ptr(),
                                 kc.data_ptr(), vt.data_ptr(), out.data_ptr(), ws.data_ptr(), qkv.shape[0], Hq, Hkv, D, S_max, int(max_len),
                                 float(D ** -0.5), 1 if packed else 0)
    assert st == 0, st


# fragment-packed cache images (decode_attn_v2.hip, PACKED): K [S/16][D/32][fq 4][frow 16][8]; V^T [D/16][S/32][fq 4][frow 16][run 2][4]
def pack_k(kc):
    B, G, S, D = kc.shape
    return kc.view(B, G, S // 16, 16, D // 32, 4, 8).permute(0, 1, 2, 4, 5, 3, 6).contiguous()


def unpack_k(kp, S, D):
    B, G = kp.shape[:2]
    return kp.view(B, G, S // 16, D // 32, 4, 16, 8).permute(0, 1, 2, 5, 3, 4, 6).reshape(B, G, S, D)


def pack_vt(vt):
    B, G, D, S = vt.shape
    return vt.view(B, G, D // 16, 16, S // 32, 2, 4, 4).permute(0, 1, 2, 4, 6, 3, 5, 7).contiguous()


def unpack_vt(vp, D, S):
    B, G = vp.shape[:2]
    return vp.view(B, G, D // 16, S // 32, 4, 16, 2, 4).permute(0, 1, 2, 5, 3, 6, 4, 7).reshape(B, G, D, S)


def rnd(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).cuda().to(H)


def parity(Hq, Hkv, slots, S_max=640, D=128, sec=(16, 24, 24), seed=1):
    B = len(slots)
    qkv = rnd(B, (Hq + 2 * Hkv) * D, seed=33 + seed)
    kc = rnd(B, Hkv, S_max, D, seed=34 + seed)
    vt = rnd(B, Hkv, S_max, D, seed=35 + seed).transpose(2, 3).contiguous()
    slot_t = torch.tensor(slots, dtype=torch.int32, device="cuda")
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float) / D))).cuda()
    gpos = torch.randint(0, 4000, (3, B), dtype=torch.int32, generator=torch.Generator().manual_seed(seed)).cuda()
    csx = torch.zeros(B, D // 2, 2, device="cuda")
    ops.rope_table(gpos, inv, csx, D, sec)
    o1, o2 = torch.zeros(B, Hq * D, device="cuda", dtype=H), torch.zeros(B, Hq * D, device="cuda", dtype=H)
    k1, v1, k2, v2c = kc.clone(), vt.clone(), kc.clone(), vt.clone()
    ops.decode_attn_rope(qkv, csx, slot_t, k1, v1, o1, ops.new_decode_workspace(B, Hkv, D, S_max, "cuda"), Hq, Hkv, D, S_max, S_max)
    v2(qkv, csx, slot_t, k2, v2c, o2, ops.new_decode_workspace(B, Hkv, D, S_max, "cuda"), Hq, Hkv, D, S_max, S_max)
    o3 = torch.zeros_like(o2)
    kp, vp = pack_k(kc), pack_vt(vt)
    assert torch.equal(unpack_k(kp, S_max, D), kc) and torch.equal(unpack_vt(vp, D, S_max), vt)
    v2(qkv, csx, slot_t, kp, vp, o3, ops.new_decode_workspace(B, Hkv, D, S_max, "cuda"), Hq, Hkv, D, S_max, S_max, packed=True)
    torch.cuda.synchronize()
    print(f"[packed caches] outputs equal the row-major v2's: {torch.equal(o3, o2)} ({(o3 != o2).sum().item()} differ); K append equal "
          f"{torch.equal(unpack_k(kp, S_max, D), k1)}, V append equal {torch.equal(unpack_vt(vp, D, S_max), v1)}")
    # fp32 statement of the same attention from the library's rotated q and its caches
    qb = torch.zeros(B, Hq * D, device="cuda", dtype=H)
    k3, v3 = kc.clone(), vt.clone()
    ops.llm_qkv_post(qkv, gpos, inv, qb, k3, v3, Hq, Hkv, D, S_max, sec, slot=slot_t)
    rep = Hq // Hkv
    ref = torch.zeros(B, Hq * D, device="cuda")
    for b in range(B):
        L = slots[b] + 1
        kk = k3[b, :, :L].float().repeat_interleave(rep, 0)
        vv = v3[b, :, :, :L].float().transpose(1, 2).repeat_interleave(rep, 0)
        sc = torch.einsum("hd,hld->hl", qb[b].float().view(Hq, D), kk) * D ** -0.5
        ref[b] = torch.einsum("hl,hld->hd", torch.softmax(sc, -1), vv).reshape(-1)
    for nw in (4, 8):                                              # v3: appends bit-identical, outputs at the same distance to fp32
        o4 = torch.zeros_like(o2)
        k4, v4 = kc.clone(), vt.clone()
        attn_v3(qkv, csx, slot_t, k4, v4, o4, Hq, Hkv, D, S_max, nw=nw)
        o5 = torch.zeros_like(o2)
        kp4, vp4 = pack_k(kc), pack_vt(vt)
        attn_v3(qkv, csx, slot_t, kp4, vp4, o5, Hq, Hkv, D, S_max, packed=True, nw=nw)
        torch.cuda.synchronize()
        print(f"[v3 nw={nw}] K append equal {torch.equal(k4, k1)}, V append equal {torch.equal(v4, v1)}; |v3 - fp32| max {(o4.float() - ref).abs().max().item():.3e} "
              f"(library {(o1.float() - ref).abs().max().item():.3e}); |v3 - library| max {(o4.float() - o1.float()).abs().max().item():.3e} "
              f"({(o4 != o1).sum().item()} of {o1.numel()} differ); packed == row-major: {torch.equal(o5, o4)}, packed appends equal "
              f"{torch.equal(unpack_k(kp4, S_max, D), k1) and torch.equal(unpack_vt(vp4, D, S_max), v1)}; finite {bool(torch.isfinite(o4.float()).all())}", flush=True)
    e1 = (o1.float() - ref).abs().max().item()
    e2 = (o2.float() - ref).abs().max().item()
    d12 = (o1.float() - o2.float()).abs().max().item()
    print(f"[parity Hq {Hq} Hkv {Hkv} slots {slots}] K append equal {torch.equal(k1, k2)}, V append equal {torch.equal(v1, v2c)}; "
          f"|library - fp32| max {e1:.3e}, |v2 - fp32| max {e2:.3e}, |library - v2| max {d12:.3e} ({(o1 != o2).sum().item()} of {o1.numel()} outputs differ); "
          f"finite {bool(torch.isfinite(o2.float()).all())}", flush=True)


def graph_time(fn, n=100):
    for _ in range(8):
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


def timing(B, S_max, slot, Hq=16, Hkv=2, D=128):
    qkv = rnd(B, (Hq + 2 * Hkv) * D, seed=5)
    nset = 4
    kcs = [rnd(B, Hkv, S_max, D, seed=6 + i) for i in range(nset)]
    vts = [rnd(B, Hkv, D, S_max, seed=16 + i) for i in range(nset)]
    cs = torch.randn(B, D // 2, 2, device="cuda")
    slot_t = torch.full((B,), slot, dtype=torch.int32, device="cuda")
    out = torch.zeros(B, Hq * D, device="cuda", dtype=H)
    ws = ops.new_decode_workspace(B, Hkv, D, S_max, "cuda")
    max_len = slot + 1
    j = [0]

    def lib():
        j[0] += 1
        ops.decode_attn_rope(qkv, cs, slot_t, kcs[j[0] % nset], vts[j[0] % nset], out, ws, Hq, Hkv, D, S_max, max_len)

    def new():
        j[0] += 1
        v2(qkv, cs, slot_t, kcs[j[0] % nset], vts[j[0] % nset], out, ws, Hq, Hkv, D, S_max, max_len)

    def newp():                                                    # same buffers read as packed images (timing only: the bytes are what matters)
        j[0] += 1
        v2(qkv, cs, slot_t, kcs[j[0] % nset], vts[j[0] % nset], out, ws, Hq, Hkv, D, S_max, max_len, packed=True)

    def mk3(nw, packed):
        def f():
            j[0] += 1
            attn_v3(qkv, cs, slot_t, kcs[j[0] % nset], vts[j[0] % nset], out, Hq, Hkv, D, S_max, packed=packed, nw=nw)
        return f
    t34, t38, t34p, t38p = graph_time(mk3(4, False)), graph_time(mk3(8, False)), graph_time(mk3(4, True)), graph_time(mk3(8, True))
    t1, t2, t3 = graph_time(lib), graph_time(new), graph_time(newp)
    kv_mb = B * Hkv * (slot + 1) * D * 2 * 2 / 1e6
    print(f"[timing B {B:3d} heads {Hq}:{Hkv} keys {slot + 1:4d}] v3 ONE launch, us per call: 4 waves {t34:6.2f} ({t1 / t34:.2f}x of the library), 8 waves {t38:6.2f} "
          f"({t1 / t38:.2f}x); on packed caches 4 waves {t34p:6.2f} ({t1 / t34p:.2f}x), 8 waves {t38p:6.2f} ({t1 / t38p:.2f}x) = {kv_mb / min(t34p, t38p):.2f} TB/s of KV", flush=True)
    print(f"[timing B {B:3d} heads {Hq}:{Hkv} keys {slot + 1:4d}] attention + merge, us per call: library {t1:6.2f}, v2 {t2:6.2f} ({t1 / t2:.2f}x), "
          f"v2 on packed caches {t3:6.2f} ({t1 / t3:.2f}x); {kv_mb:.1f} MB of KV: {kv_mb / t1:.2f} → {kv_mb / t2:.2f} → {kv_mb / t3:.2f} TB/s", flush=True)


def main():
    parity(16, 2, [0, 3, 63, 64, 333, 639])
    parity(16, 2, [577, 70, 600, 17, 130, 255], seed=2)
    parity(28, 4, [1, 62, 65, 601, 382, 96], seed=3)              # group of 7 (PaDT_Pro_7B heads): one dead row inside the first 8
    for B, S_max, slot in ((64, 640, 600), (8, 640, 600), (128, 640, 600), (128, 1344, 950)):
        timing(B, S_max, slot)
    timing(64, 640, 600, Hq=28, Hkv=4)


if __name__ == "__main__":
    main()
