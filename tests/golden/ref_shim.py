"""Import-time shims that let the UNMODIFIED reference (/root/reference) import
and run on CPU in the build container.  Used only by make_golden.py; the GPU box
has no /root/reference, so nothing at test run time imports this.

What is stubbed (all outside the hot path): torchvision (only
``<arch>(True).fc.in_features`` is read, models.py:125-126), colorama,
tensorboardX, ``.cuda()`` (identity), ``torch.cuda.device_count`` (1) and the
name ``torch`` that models.py:14 expects to leak from ``from torch.nn.init
import *``.
"""
import builtins
import sys
import types

import torch

REF = "/root/reference"
_DIMS = dict(resnet18=512, resnet34=512, resnet50=2048, resnet101=2048, resnet152=2048)


def install():
    builtins.torch = torch
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")

    def _mk(dim):
        def ctor(pretrained=True):
            return types.SimpleNamespace(fc=types.SimpleNamespace(in_features=dim))
        return ctor

    for name, dim in _DIMS.items():
        setattr(tvm, name, _mk(dim))
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm

    col = types.ModuleType("colorama")
    col.init = lambda **kw: None

    class _Blank:
        def __getattr__(self, k):
            return ""

    col.Fore = col.Back = col.Style = _Blank()
    sys.modules["colorama"] = col

    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    sys.modules["tensorboardX"] = tbx

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.device_count = lambda: 1
    if REF not in sys.path:
        sys.path.append(REF)


def fixed_accuracy(output, target, topk=(1,)):
    """main.py:809-822 with .reshape instead of .view at :820 (the reference
    line raises on torch >= 1.7 for a non-contiguous slice)."""
    maxk = max(topk)
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    res = []
    for k in topk:
        correct_k = correct[:k].reshape(-1).float().sum(0)
        res.append(correct_k.mul_(100.0 / batch_size))
    return res
