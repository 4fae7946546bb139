"""quickrank_amd -- MI355X-native LambdaMART/GBRT training and tree-ensemble scoring.

The package holds only what QuickRank's hot path needs:
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (include/qr_hip.h)
  _capi.py   ctypes binding of that C-ABI
  trainer.py host mirror of Mart / LambdaMart (mart.cc, lambdamart.cc), incl. save / load_model_from_file
  io.py      SVMLight, XML model and score files through host/'s C++ classes
  dist.py    feature-block sharding over torch.distributed (RCCL on GPUs)
There is no CPU fallback: importing is cheap, but every compute entry point
needs libqr_hip.so and a visible gfx950 device and fails loudly otherwise.
"""
from ._capi import Context, QrError, NODE_DTYPE, SPLIT_DTYPE, QR_MAX_BINS  # noqa: F401
