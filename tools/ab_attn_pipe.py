#!/usr/bin/env python3
"""A/B of the long-stream attention kernel: the plain form (lvq_tuning.attn_pipe = -1) against the software-pipelined one (attn_pipe = 1), same inputs,
outputs compared bit for bit, both timed.  Shapes as in the bench (B scenes x 12 heads x 576 queries x 4096 tiles), q hi + lo."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_vision_vqa_amd import _ffi, ops
DEV = torch.device("cuda:0"); torch.set_grad_enabled(False)
B = int(os.environ.get("SCENES", "8")); H, nq, nt, dh = 12, 576, 4096, 64
d = H * dh; hw = nt * 64
def timeit(fn, iters=3):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
def ab(name, fn, flops):
    """Outputs with the KV split pinned (same partial sums in both forms -> bit-identical), times with each form's own plan."""
    res = {}
    for mode in ("0", "1"):
        pipe = 1 if mode == "1" else -1                    # include/lvq.h: lvq_tuning.attn_pipe
        with _ffi.tuning(attn_pipe=pipe, attn_nsplit=8):
            out = fn(); torch.cuda.synchronize()
            outs = [o.clone() for o in out if o is not None] if isinstance(out, (tuple, list)) else [out.clone()]
        with _ffi.tuning(attn_pipe=pipe):
            res[mode] = (outs, timeit(fn))
    same = all(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
               for a, b in zip(res["0"][0], res["1"][0]))
    dmax = max(float((a.float() - b.float()).abs().max()) for a, b in zip(res["0"][0], res["1"][0]))
    t0, t1 = res["0"][1], res["1"][1]
    print(f"{name:40s} plain {t0:8.3f} ms ({flops / t0 / 1e9:7.1f} TF)  pipelined {t1:8.3f} ms ({flops / t1 / 1e9:7.1f} TF)  "
          f"bit-identical {same}  max|diff| {dmax:.3e}", flush=True)
    return same
q1 = ops.cast(torch.randn(nq, d, device=DEV), True)
q = (q1[0].repeat(B, 1), q1[1].repeat(B, 1))
kv = (torch.randn(hw + B * hw, 2 * d, device=DEV) * 0.7).to(torch.bfloat16)
e_idx = torch.arange(hw, device=DEV, dtype=torch.int32).view(1, hw)
g = torch.Generator(device=DEV).manual_seed(1)
dirty = torch.rand(B, hw, device=DEV, generator=g) < 0.276
order = dirty.view(B, nt, 64).permute(1, 0, 2).reshape(-1)
num = (torch.cumsum(order.int(), 0) - 1).view(nt, B, 64).permute(1, 0, 2).reshape(B, hw)
s = torch.where(dirty, hw + num, e_idx.expand(B, hw)).to(torch.int32).contiguous()
sc = 1 / math.sqrt(dh)
fl = 4.0 * B * nq * hw * d
ok = True
for prec, qq, q1q in (("q hi+lo", q, q1), ("q plain", (q[0], None), (q1[0], None))):
    ok &= ab(f"tiled stream, {prec}", lambda: ops.attention_tiled(qq, kv, s, batch=B, n_heads=H, nq=nq, n_tiles=nt, dh=dh, scale=sc), fl * (1.5 if qq[1] is not None else 1))
    dk = kv[hw:]
    ok &= ab(f"dense stream, {prec}", lambda: ops.attention(qq, (dk, None), (dk[:, d:], None), batch=B, n_heads=H, n_kv_heads=H, nq=nq, nkv=hw, dh=dh,
                                                       q_strides=(nq * d, d, dh), k_strides=(hw * 2 * d, 2 * d, dh), v_strides=(hw * 2 * d, 2 * d, dh), scale=sc),
             fl * (1.5 if qq[1] is not None else 1))
pair_src, pair_info = ops.bev_scene_pairs(s, B, nt, hw)
keys = float(pair_info.view(B, 2)[:, 0].sum()) * 64
tot = ops.attention_stream_totals(q1, kv[:hw], n_heads=H, nq=nq, nkv=hw, dh=dh, scale=sc)
ok &= ab("signed pair stream (27.6 % dirty)", lambda: ops.attention_tiled_signed(q1, kv, s, pair_src, pair_info, tot, batch=B, n_heads=H, nq=nq, n_tiles=nt, dh=dh,
                                                                                 scale=sc, shared_q=True), 4.0 * nq * keys * d * 1.5)
# short / odd stream lengths through the signed call: pair lists of 0, 1, 2, 3, 5 tiles and a > 50 % dirty scene (full list, unsigned)
for frac in (0.0, 0.0002, 0.001, 0.6):
    g2 = torch.Generator(device=DEV).manual_seed(7)
    dirty2 = torch.rand(B, hw, device=DEV, generator=g2) < frac
    order2 = dirty2.view(B, nt, 64).permute(1, 0, 2).reshape(-1)
    num2 = (torch.cumsum(order2.int(), 0) - 1).view(nt, B, 64).permute(1, 0, 2).reshape(B, hw)
    s2 = torch.where(dirty2, hw + num2, e_idx.expand(B, hw)).to(torch.int32).contiguous()
    ps2, pi2 = ops.bev_scene_pairs(s2, B, nt, hw)
    ok &= ab(f"signed, dirty fraction {frac}", lambda: ops.attention_tiled_signed(q1, kv, s2, ps2, pi2, tot, batch=B, n_heads=H, nq=nq, n_tiles=nt, dh=dh,
                                                                                 scale=sc, shared_q=True), 1.0)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
