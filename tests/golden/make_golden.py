#!/usr/bin/env python3
"""Generate golden vectors from the reference implementation (build container only).

Imports the reference's four model files BY PATH from /root/reference (they are
never copied into this repository), runs seeded inputs through them on CPU/fp32
and stores inputs + outputs as small .npz fixtures next to this script.  The
fixtures are data only: tensors, scalars and sha256 digests.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
    python tests/golden/make_golden.py temp       # only the scratch-variant models of Classification/modules/temp.py
    python tests/golden/make_golden.py more       # only the remaining classification classes and the dataset statements
"""
import hashlib
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("URESNET_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
FILES = {
    "seg": "Segmentation/Cityscapes/modules/U_ResNet.py",
    "ice": "Segmentation/Ice-Seg/modules/U_ResNet.py",
    "cifar": "Classification/modules/U_ResNet_CIFAR.py",
    "mini": "Classification/modules/U_ResNet_MiniImageNet.py",
    "temp": "Classification/modules/temp.py",
}


def load_ref(tag):
    path = os.path.join(REF, FILES[tag])
    spec = importlib.util.spec_from_file_location("ref_" + tag, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def npy(t):
    return t.detach().cpu().numpy().copy()


def randn(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def module_case(name, build, in_shapes, seed=0, eval_after=False):
    """One train-mode forward/backward of a block; optional eval-mode forward afterwards."""
    torch.manual_seed(seed)
    m = build()
    m.train()
    # make BN affine parameters and biases non-trivial so parity exercises them
    g = torch.Generator().manual_seed(seed + 17)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.2)
    rec = {}
    for k, v in m.state_dict().items():
        rec["w0/" + k] = npy(v)
    xs = [randn(s, seed + 100 + i).requires_grad_(True) for i, s in enumerate(in_shapes)]
    out = m(*xs)
    gout = randn(out.shape, seed + 200)
    (out * gout).sum().backward()
    for i, x in enumerate(xs):
        rec["x%d" % i] = npy(x)
        rec["dx%d" % i] = npy(x.grad)
    rec["out"] = npy(out)
    rec["gout"] = npy(gout)
    for k, p in m.named_parameters():
        if p.grad is not None:
            rec["grad/" + k] = npy(p.grad)
    for k, v in m.state_dict().items():
        if "running_" in k or "num_batches" in k:
            rec["w1/" + k] = npy(v)
    if eval_after:
        m.eval()
        with torch.no_grad():
            rec["out_eval"] = npy(m(*[x.detach() for x in xs]))
    np.savez(os.path.join(OUT, name + ".npz"), **rec)
    print("wrote", name, {k: v.shape for k, v in rec.items() if not k.startswith(("w0/", "grad/", "w1/"))})


def shuffle_case(seg):
    rec = {}
    x = randn((2, 16, 4, 6), 5)
    rec["x"] = npy(x)
    for g in (2, 4, 8, 16):
        rec["g%d" % g] = npy(seg.channel_shuffle(x, g))
    np.savez(os.path.join(OUT, "channel_shuffle.npz"), **rec)
    print("wrote channel_shuffle")


def loss_case():
    rec = {}
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(2, 19, 8, 16, generator=g).requires_grad_(True)
    tgt = torch.randint(0, 19, (2, 8, 16), generator=g)
    tgt[torch.rand(2, 8, 16, generator=g) < 0.1] = 255
    loss = nn.CrossEntropyLoss(ignore_index=255)(logits, tgt.long())
    loss.backward()
    rec.update(ce_logits=npy(logits), ce_target=npy(tgt), ce_loss=npy(loss), ce_dlogits=npy(logits.grad))
    # plain CE (classification)
    l2 = torch.randn(8, 10, generator=g).requires_grad_(True)
    t2 = torch.randint(0, 10, (8,), generator=g)
    loss2 = nn.CrossEntropyLoss()(l2, t2)
    loss2.backward()
    rec.update(cls_logits=npy(l2), cls_target=npy(t2), cls_loss=npy(loss2), cls_dlogits=npy(l2.grad))
    # BCE with logits
    l3 = (torch.randn(2, 1, 8, 8, generator=g) * 3).requires_grad_(True)
    t3 = (torch.rand(2, 1, 8, 8, generator=g) > 0.5).float()
    loss3 = nn.BCEWithLogitsLoss()(l3, t3)
    loss3.backward()
    rec.update(bce_logits=npy(l3), bce_target=npy(t3), bce_loss=npy(loss3), bce_dlogits=npy(l3.grad))
    # argmax with engineered ties (first maximum wins) and sigmoid>0.5 around +-0
    a = torch.randn(1, 19, 4, 8, generator=g)
    a[0, 3, 0, :] = 9.0
    a[0, 7, 0, :] = 9.0            # tie between classes 3 and 7 -> 3
    a[0, :, 1, 0] = 0.0            # all-equal -> 0
    a[0, 18, 2, :] = 50.0
    rec.update(am_logits=npy(a), am_mask=npy(a.argmax(1)))
    s = torch.tensor([0.0, -0.0, 1e-45, -1e-45, 1e-8, -1e-8, 3.0, -3.0, float("inf"), -float("inf")]).view(1, 1, 2, 5)
    rec.update(sg_logits=npy(s), sg_mask=npy(torch.sigmoid(s) > 0.5))
    np.savez(os.path.join(OUT, "losses_masks.npz"), **rec)
    print("wrote losses_masks")


def metrics_case():
    """calculate_metrics (train_UResNetAblitionStudy.py:70-100).  The script itself cannot be imported (it needs
    albumentations / the dataset package at import time), so the function's definition is compiled on its own from the
    script's syntax tree and run with the sklearn functions it names."""
    import ast
    from sklearn.metrics import accuracy_score, f1_score, jaccard_score, precision_score, recall_score
    path = os.path.join(REF, "Segmentation/Cityscapes/train_UResNetAblitionStudy.py")
    tree = ast.parse(open(path, encoding="utf-8").read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "calculate_metrics"]
    ns = dict(precision_score=precision_score, recall_score=recall_score, accuracy_score=accuracy_score,
              f1_score=f1_score, jaccard_score=jaccard_score)
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    ref_metrics = ns["calculate_metrics"]
    keys = ("precision", "recall", "accuracy", "f1", "iou", "dice")
    g = torch.Generator().manual_seed(21)
    rec = {}
    # 0: all 19 classes; 1: targets use 6 classes, predictions 8 (absent labels, zero divisions); 2: one pixel class only
    for i in range(3):
        logits = torch.randn(2, 19, 16, 32, generator=g)
        tgt = torch.randint(0, 19, (2, 16, 32), generator=g)
        if i == 1:
            logits[:, 8:] -= 50.0
            tgt = torch.randint(0, 6, (2, 16, 32), generator=g)
        if i == 2:
            tgt = torch.full((2, 16, 32), 4)
            logits[:, 4] += 2.0
        tgt[torch.rand(2, 16, 32, generator=g) < 0.1] = 255
        m = ref_metrics(logits, tgt, 19)
        rec["logits%d" % i] = npy(logits)
        rec["target%d" % i] = npy(tgt)
        rec["metrics%d" % i] = np.array([m[k] for k in keys], dtype=np.float64)
    np.savez(os.path.join(OUT, "seg_metrics.npz"), **rec)
    print("wrote seg_metrics")


def dataset_case():
    """Label remap and the no-transform image conversion of CityscapesSemantic (DataInstance/Cityscapes.py:191-194,
    220-223, 238, 242).  The dataset module needs PIL / albumentations at import time and files on disk, so exactly those
    statements are compiled on their own from the file's syntax tree and run on synthetic label-id and image arrays."""
    import ast
    import types
    path = os.path.join(REF, "Segmentation/Cityscapes/DataInstance/Cityscapes.py")
    tree = ast.parse(open(path, encoding="utf-8").read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CityscapesSemantic"][0]
    fns = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}

    def stmts(fn, lo, hi):
        out = []

        def walk(body):
            for n in body:
                if isinstance(n, (ast.Assign, ast.For)) and lo <= n.lineno and n.end_lineno <= hi:
                    out.append(n)
                else:
                    for field in ("body", "orelse"):
                        walk(getattr(n, field, []) or [])
        walk(fn.body)
        return out

    table = stmts(fns["__init__"], 191, 194)            # valid_classes, class_map, ignore_index
    remap = stmts(fns["__getitem__"], 220, 223)         # mask = 255; mask[label == k] = class_map[k]; semantic_label = mask
    plain = stmts(fns["__getitem__"], 238, 238) + stmts(fns["__getitem__"], 242, 242)    # /255.0 image, .long() label
    assert len(table) == 3 and len(remap) == 3 and len(plain) == 2, (len(table), len(remap), len(plain))
    self = types.SimpleNamespace()
    exec(compile(ast.Module(body=table, type_ignores=[]), path, "exec"), {"self": self})
    rng = np.random.RandomState(31)
    ids = rng.randint(0, 34, size=(2, 24, 40)).astype(np.uint8)       # raw Cityscapes ids 0..33
    ids[0, :2] = 255                                                  # and values outside the id range
    ids[1, 5, :7] = 200
    img = rng.randint(0, 256, size=(2, 24, 40, 3)).astype(np.uint8)
    rec = {"ids": ids, "image": img, "valid_classes": np.array(self.valid_classes), "ignore_index": np.array(self.ignore_index)}
    targets, images = [], []
    for i in range(2):
        ns = {"self": self, "np": np, "torch": torch, "semantic_label": ids[i], "image": img[i]}
        exec(compile(ast.Module(body=remap + plain, type_ignores=[]), path, "exec"), ns)
        targets.append(npy(ns["semantic_label"]))
        images.append(npy(ns["image"]))
    rec["target"] = np.stack(targets)          # int64, 19 classes + 255
    rec["image_div255"] = np.stack(images)     # CHW float32
    np.savez(os.path.join(OUT, "cityscapes_dataset.npz"), **rec)
    print("wrote cityscapes_dataset", rec["target"].dtype, rec["image_div255"].shape)


SELECTED = ("conv1.0.weight", "ResLayer2.0.shortcut.0.weight", "ResLayer4.1.left.3.weight", "Upsample.pwc.0.weight",
            "in_conv.0.weight", "down2.1.3.weight", "up1.up.weight", "up1.up.bias", "up4.conv.0.weight",
            "out_conv.0.weight", "out_conv.0.bias", "out_conv.2.weight", "out_conv.2.bias", "down4.1.1.weight")


def seg_model_case(name, build, classes, shape, steps=3):
    torch.manual_seed(0)
    m = build()
    m.train()
    rec = {}
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    rec["keys"] = np.array(list(sd0.keys()))
    rec["w0_sha"] = np.array([sha(v) for v in sd0.values()])
    x = randn(shape, 1234)
    g = torch.Generator().manual_seed(4321)
    if classes > 1:
        tgt = torch.randint(0, classes, (shape[0], shape[2], shape[3]), generator=g)
        tgt[torch.rand(tgt.shape, generator=g) < 0.1] = 255
        lossf = lambda o: nn.CrossEntropyLoss(ignore_index=255)(o, tgt.long())
    else:
        tgt = (torch.rand(shape[0], 1, shape[2], shape[3], generator=g) > 0.5).float()
        lossf = lambda o: nn.BCEWithLogitsLoss()(o, tgt)
    rec["x"], rec["target"] = npy(x), npy(tgt)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    losses = []
    for s in range(steps):
        out = m(x)
        loss = lossf(out)
        opt.zero_grad()
        loss.backward()
        if s == 0:
            rec["logits"] = npy(out)
            names = [k for k, p in m.named_parameters() if p.grad is not None]
            rec["grad_keys"] = np.array(names)
            rec["grad_norm"] = np.array([p.grad.double().norm().item() for k, p in m.named_parameters() if p.grad is not None])
            rec["grad_sha"] = np.array([sha(p.grad) for k, p in m.named_parameters() if p.grad is not None])
            for k, p in m.named_parameters():
                if k in SELECTED:
                    rec["grad/" + k] = npy(p.grad)
        opt.step()
        losses.append(loss.item())
        if s == 0:
            rec["w1_sha"] = np.array([sha(v) for v in m.state_dict().values()])
            rec["w1_norm"] = np.array([v.double().norm().item() for v in m.state_dict().values()])
    rec["losses"] = np.array(losses, dtype=np.float64)
    rec["wN_norm"] = np.array([v.double().norm().item() for v in m.state_dict().values()])
    rec["wN_sha"] = np.array([sha(v) for v in m.state_dict().values()])
    m.eval()
    with torch.no_grad():
        oe = m(x)
    rec["logits_eval"] = npy(oe)
    rec["mask_eval"] = npy(oe.argmax(1) if classes > 1 else torch.sigmoid(oe) > 0.5)
    np.savez(os.path.join(OUT, name + ".npz"), **rec)
    print("wrote", name, "losses", losses)


def cls_model_case(name, build, nclass, shape, lr, steps=3):
    torch.manual_seed(0)
    m = build()
    m.train()
    rec = {}
    rec["keys"] = np.array(list(m.state_dict().keys()))
    rec["w0_sha"] = np.array([sha(v) for v in m.state_dict().values()])
    x = randn(shape, 1234)
    g = torch.Generator().manual_seed(4321)
    y = torch.randint(0, nclass, (shape[0],), generator=g)
    rec["x"], rec["target"] = npy(x), npy(y)
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    losses = []
    for s in range(steps):
        opt.zero_grad()
        out = m(x)
        loss = nn.CrossEntropyLoss()(out, y)
        loss.backward()
        if s == 0:
            rec["logits"] = npy(out)
            names = [k for k, p in m.named_parameters() if p.grad is not None]
            rec["grad_keys"] = np.array(names)
            rec["grad_norm"] = np.array([p.grad.double().norm().item() for k, p in m.named_parameters() if p.grad is not None])
            rec["nograd_keys"] = np.array([k for k, p in m.named_parameters() if p.grad is None])
        opt.step()
        losses.append(loss.item())
    rec["losses"] = np.array(losses, dtype=np.float64)
    rec["wN_norm"] = np.array([v.double().norm().item() for v in m.state_dict().values()])
    np.savez(os.path.join(OUT, name + ".npz"), **rec)
    print("wrote", name, "losses", losses)


def main():
    torch.set_num_threads(8)
    if "more" in sys.argv[1:]:
        return main_more()
    if "temp" not in sys.argv[1:]:
        main_families()
    temp = load_ref("temp")
    only = [a for a in sys.argv[1:] if a != "temp"]
    global cls_model_case
    every = cls_model_case
    cls_model_case = lambda name, *a, **k: every(name, *a, **k) if not only or name in only else None
    cls_model_case("temp_nlnet", lambda: temp.NlNet(100, 3), 100, (2, 3, 224, 224), 0.01, steps=2)
    cls_model_case("temp_uresnet2", lambda: temp.UResNet2(100, 3), 100, (2, 3, 224, 224), 0.01, steps=1)
    cls_model_case("temp_eresnet", lambda: temp.E_ResNet(100, 3), 100, (2, 3, 224, 224), 0.01, steps=1)
    cls_model_case("temp_uresnet", lambda: temp.U_ResNet(100, 3), 100, (2, 3, 224, 224), 0.01, steps=1)


def main_families():
    seg, ice, cifar, mini = (load_ref(t) for t in ("seg", "ice", "cifar", "mini"))
    module_case("basicblock_64_64", lambda: seg.BasicBlock(64, 64), [(2, 64, 8, 32)], eval_after=True)
    module_case("basicblock_64_128", lambda: seg.BasicBlock(64, 128), [(2, 64, 8, 32)])
    module_case("basicblock_64_128_s2", lambda: cifar.BasicBlock(64, 128, stride=2), [(2, 64, 8, 16)])
    module_case("doubleconv_3_16", lambda: seg.DoubleConv(3, 16), [(2, 3, 16, 32)], eval_after=True)
    module_case("doubleconv_64_64", lambda: seg.DoubleConv(64, 64), [(2, 64, 16, 32)])
    module_case("down_16_32", lambda: seg.Down(16, 32), [(2, 16, 16, 32)])
    module_case("up_64_32_even", lambda: seg.Up(64, 32), [(2, 64, 8, 16), (2, 32, 16, 32)])
    module_case("up_64_32_odd", lambda: seg.Up(64, 32), [(2, 64, 8, 16), (2, 32, 17, 35)])
    module_case("eucb_32_16", lambda: seg.EUCB(32, 16), [(2, 32, 8, 16)], eval_after=True)
    module_case("eucb2_32_16", lambda: seg.EUCB2(32, 16), [(2, 32, 8, 16)], eval_after=True)
    module_case("eucb2x3_32_16_s3", lambda: cifar.EUCB2(32, 16, scale_factor=3), [(2, 32, 4, 4)])
    module_case("outconv_seg_80_19", lambda: seg.OutConv(80, 19), [(1, 80, 16, 32)], eval_after=True)
    module_case("outconv_cls_128_100", lambda: cifar.OutConv(128, 100), [(2, 128, 4, 4)])
    shuffle_case(seg)
    loss_case()
    metrics_case()
    seg_model_case("seg_newEUCB_c8", lambda: seg.UResNetNewEUCB(19, 3, 8), 19, (1, 3, 32, 64))
    seg_model_case("seg_uresnet_c8_bce", lambda: ice.UResNet(1, 3, 8), 1, (2, 3, 32, 32))
    seg_model_case("seg_noEUCB_c8", lambda: seg.UResNetNoEUCB(19, 3, 8), 19, (1, 3, 32, 48), steps=1)
    seg_model_case("ice_noresblock_c8", lambda: ice.UResNetNoResBlock(1, 3, 8), 1, (1, 3, 32, 32), steps=1)
    cls_model_case("cifar_uresnet4_nosu", lambda: cifar.UResNet4_No_SU(10, 3, scale=3), 10, (4, 3, 32, 32), 0.01)
    main_more(cifar, mini)


def broken_case(cifar):
    """The three classes of U_ResNet_CIFAR.py that cannot run at 32x32 as committed (:296, :342, :389): parameter names,
    seeded initialisation, and the exception the reference's forward raises."""
    import contextlib
    import io
    rec = {}
    for name, build in (("U_ResNet", lambda: cifar.U_ResNet()), ("S_ResNet", lambda: cifar.S_ResNet()),
                        ("NlNet", lambda: cifar.NlNet(10, 3))):
        torch.manual_seed(0)
        m = build().train()
        rec[name + "/keys"] = np.array(list(m.state_dict().keys()))
        rec[name + "/w0_sha"] = np.array([sha(v) for v in m.state_dict().values()])
        try:
            with contextlib.redirect_stdout(io.StringIO()):          # NlNet prints its shapes before failing
                m(randn((4, 3, 32, 32), 1234))
            raise SystemExit(name + " unexpectedly ran")
        except RuntimeError as e:
            rec[name + "/error"] = np.array(str(e).splitlines()[0])
    np.savez(os.path.join(OUT, "cifar_broken_classes.npz"), **rec)
    print("wrote cifar_broken_classes", {k: str(v) for k, v in rec.items() if k.endswith("/error")})


def main_more(cifar=None, mini=None):
    """The other classification classes that run as committed, and the dataset statements in front of the hot path."""
    cifar = cifar or load_ref("cifar")
    mini = mini or load_ref("mini")
    cls_model_case("cifar_uresnet3_noeucb", lambda: cifar.UResNet3_No_EUCB(10, 3, scale=3), 10, (4, 3, 32, 32), 0.01)
    cls_model_case("cifar_uresnet3_noublock", lambda: cifar.UResNet3_No_UBlock(10, 3, scale=3), 10, (4, 3, 32, 32), 0.01)
    cls_model_case("cifar_eresnet", lambda: cifar.E_ResNet(), 10, (4, 3, 32, 32), 0.01)
    cls_model_case("mini_uresnet", lambda: mini.U_ResNet(100, 3), 100, (2, 3, 224, 224), 0.1, steps=2)
    dataset_case()
    broken_case(cifar)


if __name__ == "__main__":
    sys.exit(main())
