"""lidar-vision-vqa_amd -- MI355X-native LiDAR+vision fusion hot path.

Import name: `lidar_vision_vqa_amd` (the root-level shim `lidar_vision_vqa_amd.py` maps it onto this
directory, whose on-disk name carries a hyphen).  Sub-packages:

  lidar   voxeliser / VFE / BEV-scatter plugins with the OpenPCDet `batch_dict` protocol
  fusion  VATBlock / VATLiDAR / VATVision / VisionAdapter with the reference's nn.Module API
  head    prefix assembly + stand-in decoder head
  dist    scene sharding + fused all-reduce over RCCL
  _ffi    ctypes binding of the C-ABI library built from csrc/ (include/lvq.h)
  synth   synthetic scenes / weights (no dataset or checkpoint is reachable offline)
"""
__version__ = "0.1.0"
