import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from lidar_vision_vqa_amd import lidar, synth
from oracle import lidar_oracle as LO
DEV = "cuda:0"
RNG = list(synth.PC_RANGE_NUSC)
nsc = int(sys.argv[1]) if len(sys.argv) > 1 else 16
base = [synth.scene_points("C", 120000, 1100 + i) for i in range(4)]
scenes = (base * 4)[:nsc]
pts = torch.from_numpy(np.concatenate(scenes)).to(DEV)
off = torch.tensor(np.concatenate(([0], np.cumsum([len(x) for x in scenes]))), dtype=torch.int32, device=DEV)
gen = lidar.VoxelGeneratorWrapper(synth.VOXEL_01, RNG, 4, 10, 160000)
res = {}
for name, env in [("hashed", None), ("binned", "LVQ_VOXEL_BINNED")]:
    if env: os.environ[env] = "1"
    out = gen.generate_batch_device(pts, off, nsc)
    torch.cuda.synchronize()
    res[name] = [t.cpu().numpy() for t in out]
    if env: del os.environ[env]
exp = [LO.VoxelGenerator(synth.VOXEL_01, RNG, 4, 10, 160000).generate(s) for s in base]
for name, (vox, co, num, svo) in res.items():
    print(name, "svo", svo[:6], "...")
    for s in range(nsc):
        v, c, k = exp[s % 4]
        a, b = svo[s], svo[s + 1]
        okc = (b - a == len(c)) and np.array_equal(co[a:b, 1:], c)
        okn = (b - a == len(c)) and np.array_equal(num[a:b], k)
        okv = (b - a == len(c)) and np.array_equal(vox[a:b].view(np.uint32), v.view(np.uint32))
        if not (okc and okn and okv):
            print("  scene", s, "coords", okc, "num", okn, "vox", okv, "M", b - a, "exp", len(c))
            if b - a == len(c):
                bad = np.nonzero((co[a:b, 1:] != c).any(1))[0]
                print("   first bad coord rows", bad[:5], "count", len(bad))
                if len(bad): print("   got", co[a + bad[0]], "exp", c[bad[0]])
                badn = np.nonzero(num[a:b] != k)[0]
                print("   bad num rows", badn[:5], len(badn))
