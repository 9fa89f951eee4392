import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_vision_vqa_amd import ops, pipeline as P
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = P.PipelineConfig()
pipe = P.FusionPipeline(cfg, dev, precision="bf16")
pts, off, patches, _, _ = P.synthetic_batch(cfg, 4, 1100, dev)
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
def run():
    S = 4
    mark("start")
    vox3, co3, num3, svo3 = pipe.gen3d.generate_batch_device(pts, off, S); mark("vox3d")
    feat3 = pipe.mean_vfe.forward_device(vox3, num3, svo3[S:]); mark("meanvfe")
    voxp, cop, nump, svop = pipe.genp.generate_batch_device(pts, off, S); mark("voxp")
    pf = pipe.pillar_vfe.forward_device(voxp, nump, cop, svop[S:]); mark("pillarvfe")
    bev = pipe.scatter.forward_device(pf, cop, S, svop[S:]); mark("scatter")
    x = pipe.vat_lidar.bev_tokens(bev); mark("bev_tokens")
    lt = pipe.vat_lidar(bev); mark("vat_lidar_total(incl tokens again)")
    fused = pipe.fuse(lt, patches); mark("fuse")
for mode in ("free", "sync"):
    for _ in range(3): run()
    torch.cuda.synchronize(); marks.clear()
    t0 = time.perf_counter()
    for i in range(10):
        run()
        if mode == "sync": torch.cuda.synchronize()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 100
    acc = {}
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        if n1 == "start": continue
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1) / 10
    print(mode, "wall ms/step", round(wall, 2), {k: round(v, 3) for k, v in acc.items()})
    marks.clear()
