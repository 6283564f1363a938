#!/usr/bin/env python
"""Run the reference's OWN program files - main.py (training) or test_models.py (testing), unmodified - on the MI355X path.

    python compat/run_reference.py /path/to/TA3N/main.py <the reference's command line>
    python compat/run_reference.py /path/to/TA3N/test_models.py <the reference's command line>

What it arranges, and nothing else:
 * sys.path = [compat/, this repository, ..., the TA3N checkout]: the reference's `from models import VideoModel`,
   `from loss import *`, `from opts import parser`, `from dataset import TSNDataSet`, `from utils.utils import ...`
   (main.py:12-16) resolve to the HIP-backed modules; the program file itself comes from the checkout;
 * stand-ins for the two third-party packages the reference imports and this image does not have (colorama: colour codes
   -> '', tensorboardX: SummaryWriter that swallows the calls; main.py:19-22) - only when they are not installed;
 * ta3n_amd.accel BEFORE the program's `from torch.nn.utils import clip_grad_norm_` (main.py:10), so the loop's clip and
   optimiser step run as passes over the model's flat buffers (installed from compat/models.py the rebinding came
   one import too late for main.py's own name);
 * main.py:820 `correct[:k].view(-1)` raises on torch >= 1.7 (non-contiguous slice): `accuracy` is replaced by the same
   function with `.reshape(-1)` (SURVEY.md 8b "needed patches on modern PyTorch"; the file is not touched).
"""
import importlib.util
import os
import sys
import types


def _third_party_standins() -> None:
    try:
        import colorama  # noqa: F401
    except ImportError:
        col = types.ModuleType("colorama")
        col.init = lambda **k: None

        class _Codes:
            def __getattr__(self, k):
                return ""
        col.Fore = col.Back = col.Style = _Codes()
        sys.modules["colorama"] = col
    try:
        import tensorboardX  # noqa: F401
    except ImportError:
        tbx = types.ModuleType("tensorboardX")

        class SummaryWriter:
            def __init__(self, *a, **k):
                pass

            def __getattr__(self, k):
                return lambda *a, **kw: None
        tbx.SummaryWriter = SummaryWriter
        sys.modules["tensorboardX"] = tbx


def _accuracy(output, target, topk=(1,)):
    """main.py:809-822 with .reshape(-1) at :820."""
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


def load(program: str, run_name: str = "main"):
    """Execute the reference program file as module `run_name` (its top-level code runs; its `if __name__ == '__main__'` does
    not) and return the module."""
    program = os.path.abspath(program)
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, here):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)                      # compat/ first, then the repository
    ref_dir = os.path.dirname(program)
    if ref_dir not in sys.path:
        sys.path.append(ref_dir)                   # the checkout LAST: only what compat/ does not provide comes from there
    _third_party_standins()
    from ta3n_amd import accel
    accel.install()                                # before the program binds torch.nn.utils.clip_grad_norm_ by name
    spec = importlib.util.spec_from_file_location(run_name, program)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[run_name] = mod
    spec.loader.exec_module(mod)
    if hasattr(mod, "accuracy"):
        mod.accuracy = _accuracy
    return mod


def main() -> None:
    if len(sys.argv) < 2 or not os.path.isfile(sys.argv[1]):
        raise SystemExit(__doc__)
    program = sys.argv[1]
    sys.argv = [program] + sys.argv[2:]
    name = os.path.splitext(os.path.basename(program))[0]
    if name == "main":
        load(program, "main").main()
        return
    # test_models.py is a script (no main()): its top-level code is the program
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [here, root]
    sys.path.append(os.path.dirname(os.path.abspath(program)))
    _third_party_standins()
    import runpy
    runpy.run_path(program, run_name="__main__")


if __name__ == "__main__":
    main()
