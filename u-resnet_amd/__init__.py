"""u-resnet_amd: the U-ResNet convolutional forward/backward hot path on MI355X (gfx950).

Layout
  csrc/        hand-written HIP kernels + the C ABI (include/uresnet_hip.h) -> lib/liburesnet_hip.so
  _lib.py      ctypes binding (fails loudly when the library is missing)
  ops.py       autograd operators over the C ABI (NHWC activations, fp32 parameters)
  nn.py        torch.nn-compatible leaf layers whose forwards call ops
  modules/     the reference's model files, class for class
  train.py     losses, masks, multi-tensor Adam/SGD, training steps
  ddp.py       one-process-per-GPU data parallelism (bucketed RCCL all-reduce)

The directory name contains a hyphen; import it as `uresnet_amd` (the alias module at the
repository root) or with importlib.import_module("u-resnet_amd").
"""
from . import _lib, ops, nn, train, ddp, data      # noqa: F401
from .modules import U_ResNet, U_ResNet_CIFAR, U_ResNet_MiniImageNet   # noqa: F401
from .nn import set_default_dtype, get_default_dtype   # noqa: F401

__version__ = "0.1.0"
