import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_vision_vqa_amd import ops, pipeline as P
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = P.PipelineConfig()
pipe = P.FusionPipeline(cfg, dev, precision="bf16")
pts, off, patches, _, _ = P.synthetic_batch(cfg, 4, 1100, dev)
ops.EVENTS = {}
ts = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe(pts, off, patches)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ev = ops.EVENTS; ops.EVENTS = None
print("step wall ms:", [round(t, 2) for t in ts])
for k, v in ev.items():
    print(k, [round(a.elapsed_time(b), 3) for a, b in v])

ops.EVENTS = None
for mode in ("free", "sync", "free+reduce"):
    from lidar_vision_vqa_amd import dist as D
    red = torch.zeros(cfg.d_model + 1, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        out = pipe(pts, off, patches)
        if mode == "sync": torch.cuda.synchronize()
        if mode == "free+reduce": D.reduce_step(out["fused"], red)
    torch.cuda.synchronize()
    print(mode, round((time.perf_counter() - t0) * 100, 3), "ms/step", "alloc GB", round(torch.cuda.max_memory_allocated() / 1e9, 2), "reserved", round(torch.cuda.memory_reserved() / 1e9, 2))
