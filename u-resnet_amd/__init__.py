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
import os as _os
import sys as _sys
if "torch" not in _sys.modules:
    # read by the HIP runtime when it is loaded (at `import torch`): kernel arguments in device memory shorten the
    # dispatch of the many five-microsecond kernels of a training step (0.2-0.3 ms of 28).  No effect if torch is
    # already imported; set HIP_FORCE_DEV_KERNARG=0 in the environment to keep the runtime's default.
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from . import _lib, ops, nn, train, ddp, data      # noqa: F401
from .modules import U_ResNet, U_ResNet_CIFAR, U_ResNet_MiniImageNet   # noqa: F401
from .nn import set_default_dtype, get_default_dtype   # noqa: F401

__version__ = "0.1.0"
