"""Block-level parity cases shared by the oracle-vs-golden and HIP-vs-oracle tests.

Each entry: golden file name -> builder taking a module namespace `ns` that exposes the
reference's class names (the oracle, or the HIP package's module files) and the flavour
("seg" or "cls") of the file the class comes from.
"""

MODULE_CASES = {
    "basicblock_64_64": ("seg", lambda ns: ns.BasicBlock(64, 64)),
    "basicblock_64_128": ("seg", lambda ns: ns.BasicBlock(64, 128)),
    "basicblock_64_128_s2": ("cls", lambda ns: ns.BasicBlock(64, 128, stride=2)),
    "doubleconv_3_16": ("seg", lambda ns: ns.DoubleConv(3, 16)),
    "doubleconv_64_64": ("seg", lambda ns: ns.DoubleConv(64, 64)),
    "down_16_32": ("seg", lambda ns: ns.Down(16, 32)),
    "up_64_32_even": ("seg", lambda ns: ns.Up(64, 32)),
    "up_64_32_odd": ("seg", lambda ns: ns.Up(64, 32)),
    "eucb_32_16": ("seg", lambda ns: ns.EUCB(32, 16)),
    "eucb2_32_16": ("seg", lambda ns: ns.EUCB2(32, 16)),
    "eucb2x3_32_16_s3": ("cls", lambda ns: ns.EUCB2(32, 16, scale_factor=3)),
    "outconv_seg_80_19": ("seg", lambda ns: ns.OutConv(80, 19)),
    "outconv_cls_128_100": ("cls", lambda ns: ns.OutConv(128, 100)),
}

SEG_MODEL_CASES = {
    # golden name: (class name, ctor args, number of classes)
    "seg_newEUCB_c8": ("UResNetNewEUCB", (19, 3, 8), 19),
    "seg_uresnet_c8_bce": ("UResNet", (1, 3, 8), 1),
    "seg_noEUCB_c8": ("UResNetNoEUCB", (19, 3, 8), 19),
    "ice_noresblock_c8": ("UResNetNoResBlock", (1, 3, 8), 1),
}


class OracleNS:
    """Maps the reference's per-file class names onto the oracle's (which disambiguates
    the two EUCB2 / OutConv flavours by name)."""

    def __init__(self, orc, flavour):
        self._o, self._f = orc, flavour

    def __getattr__(self, k):
        if self._f == "cls":
            k = {"EUCB2": "EUCB2x3", "OutConv": "OutConvPlain", "UNet": "ClsUNet"}.get(k, k)
        return getattr(self._o, k)
