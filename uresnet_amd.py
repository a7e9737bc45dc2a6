"""Importable alias of the `u-resnet_amd/` package (a hyphen cannot appear in an import statement).

`import uresnet_amd` returns the real package object, and every sub-module is registered under
both names, so `from uresnet_amd.modules.U_ResNet import UResNetNewEUCB` works and refers to
the same module objects as importlib.import_module("u-resnet_amd.modules.U_ResNet").
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("u-resnet_amd")
for _name, _mod in list(sys.modules.items()):
    if _name == "u-resnet_amd" or _name.startswith("u-resnet_amd."):
        sys.modules["uresnet_amd" + _name[len("u-resnet_amd"):]] = _mod
