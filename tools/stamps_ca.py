#!/usr/bin/env python3
"""Phase timeline of k_ca_fused from in-kernel s_memrealtime stamps (100 MHz): python tools/stamps_ca.py [B nq nkv]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidar_vision_vqa_amd import _ffi, fusion, synth
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
B, nq, nkv = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 32768, 196)
d, h = 768, 12
blk = fusion.VATBlock(d, h, 4 * d, 0.1).to(dev).eval()
synth.load_seeded(blk, 401)
blk.precision = "mixed"
q, kv = torch.randn(B, nq, d, device=dev), torch.randn(B, nkv, d, device=dev)
for _ in range(int(os.environ.get('WARM', '3'))):
    blk.cross_attention(q, kv)
nwg = B * ((nq + 127) // 128)
st = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
with _ffi.tuning(ca_fused_stamps=st.data_ptr()):          # include/lvq.h: lvq_tuning.ca_fused_stamps
    blk.cross_attention(q, kv)
    torch.cuda.synchronize()
raw = st.cpu().numpy().reshape(nwg, 4, 8)
t = raw[:, :, :6].astype(np.float64) / 100.0   # us
cyc = (raw[:, :, 7] - raw[:, :, 6]).astype(np.float64)
print(f"phase A: {np.median(cyc):.0f} shader cycles in {np.median(t[:, :, 3] - t[:, :, 2]):.2f} us -> clock {np.median(cyc / (t[:, :, 3] - t[:, :, 2])) / 1e3:.3f} GHz")
t0 = t[:, :, 0].min()
names = ["start", "LN done", "ring primed", "A done", "B done", "C done"]
print("per-phase duration (us), median / min / max over all waves:")
for i in range(1, 6):
    dlt = t[:, :, i] - t[:, :, i - 1]
    print(f"  {names[i]:12s} {np.median(dlt):8.2f} {dlt.min():8.2f} {dlt.max():8.2f}")
print(f"kernel span: first start -> last end {t[:, :, 5].max() - t0:.2f} us; start skew {t[:, :, 0].max() - t0:.2f} us; per-wave total median {np.median(t[:, :, 5] - t[:, :, 0]):.2f}")
