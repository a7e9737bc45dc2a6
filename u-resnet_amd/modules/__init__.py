"""Model files mirroring the reference's module files one to one."""
from . import U_ResNet, U_ResNet_CIFAR, U_ResNet_MiniImageNet, temp   # noqa: F401
