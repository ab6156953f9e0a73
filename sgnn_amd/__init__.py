"""sgnn_amd — MI355X-native sparse generative 3D convolution hot path for SG-NN.

Package contents (only what the hot path of SURVEY.md §8 needs):
  csrc/        hand-written gfx950 HIP kernels + the C ABI (include/sgnn_hip.h)
  _lib.py      ctypes binding (fails loudly without the .so / a GPU; no CPU fallback)
  scn/         `sparseconvnet`-compatible operator surface (torch/model.py:7)
  model.py     GenModel counterpart (torch/model.py:276) on the fused device-side glue
  loss.py      compute_targets / compute_loss (torch/loss.py:15-199)
  train.py     one training step + data-parallel gradient all-reduce (torch/train.py:245-268)
  synth.py     synthetic TSDF blocks in the layout scene_dataloader.collate emits
"""
__version__ = '0.1.0'
