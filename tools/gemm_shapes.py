#!/usr/bin/env python3
"""Every lvq_gemm_bf16-family call of one pipeline step (cfg-2, 32 scenes): shape, operand form, epilogue, time (one event pair each, so
short kernels include the event packets) -> the table DESIGN section 4 quotes.  python tools/gemm_shapes.py [precision]"""
import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidar_vision_vqa_amd import ops, pipeline as P
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
prec = sys.argv[1] if len(sys.argv) > 1 else "mixed"
cfg = P.PipelineConfig()
pipe = P.FusionPipeline(cfg, dev, precision=prec)
pts, off, patches, _, _ = P.synthetic_batch(cfg, 32, 1100, dev)
for _ in range(3):
    pipe(pts, off, patches)
torch.cuda.synchronize()
log = []


def wrap(name):
    fn = getattr(ops, name)

    def w(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a, **k)
        e.record()
        A, W = a[0], a[1]
        m, kk = A[0].shape
        rows = k.get("w_rows") or (a[4] if name == "linear_live_rows" else None) or (0, W[0].shape[0])
        n = rows[1] - rows[0]
        form = ("a2" if A[1] is not None else "a1") + ("w2" if W[1] is not None else "w1")
        epi = "+".join(x for x in ("gelu" if k.get("gelu") else "", "res" if k.get("residual") is not None else "", "tab" if k.get("rowtab") is not None else "",
                                   "f32" if k.get("out_f32") else "", "bf" if k.get("out_bf") else "") if x)
        log.append((name, m, n, kk, form, epi, k.get("tag"), s, e))
        return r
    setattr(ops, name, w)


for nm in ("linear", "linear_ln", "linear_live_rows"):
    if hasattr(ops, nm):
        wrap(nm)
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
pipe(pts, off, patches)
t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, m, n, k, form, epi, tag, s, e in log:
    key = (name, m, n, k, form, epi)
    c = agg.setdefault(key, [0, 0.0])
    c[0] += 1
    c[1] += s.elapsed_time(e)
tot = 0.0
print(f"step {t0.elapsed_time(t1):.2f} ms ({prec}); GEMM-family calls:")
for (name, m, n, k, form, epi), (cnt, ms) in agg.items():
    passes = {"a1w1": 1, "a1w2": 2, "a2w2": 3, "a2w1": 2}[form]
    fl = 2.0 * m * n * k * cnt
    tot += ms
    print(f"{name:17s} m={m:7d} n={n:5d} k={k:5d} {form} {epi:14s} x{cnt}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s algorithmic, {fl * passes / ms / 1e9:7.1f} executed")
print(f"total {tot:.3f} ms")
