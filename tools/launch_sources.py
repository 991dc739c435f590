"""Where the ATen launches of the steady-state prompted clip come from: every ATen operator call of one clip (TorchDispatchMode)
attributed to the innermost univs_amd source line on the Python stack (the hand-written HIP operators go through ctypes and are
not counted here).   python tools/launch_sources.py [--first]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from univs_amd import workloads as cases
dev = torch.device("cuda:0")
swin, head = cases.build_model(dev)
case = dict(cases.CFG2, H=736, W=1280)
x = cases.preprocess(cases.cfg2_frames()).to(dev)
first = "--first" in sys.argv
tv0 = (cases.targets_first_clip(case) if first else cases.targets_with_entities(case, first_frame_idx=1, n_ent=10))[0]
tvd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tv0.items()}
mk = lambda: [dict(tvd)]
NOLAUNCH = ("view", "reshape", "expand", "permute", "transpose", "t.default", "slice", "select", "unsqueeze", "squeeze", "detach", "alias",
            "as_strided", "unbind", "split", "chunk", "_unsafe_view", "empty", "sym_", "size", "stride", "is_", "unflatten", "narrow",
            "movedim", "flatten", "lift_fresh", "_local_scalar_dense", "item", "resize_", "set_", "zeros.default", "result_type", "unfold")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.src = collections.Counter()
        self.names = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(p) or ("." + p) in name for p in NOLAUNCH):
            f = sys._getframe(1)
            where = "?"
            while f is not None:
                fn = f.f_code.co_filename
                if "univs_amd/" in fn:
                    where = fn.split("univs_amd/")[-1] + ":" + str(f.f_lineno) + " " + f.f_code.co_name
                    break
                f = f.f_back
            self.src[where] += 1
            self.names[where][name] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad():
    feats = swin(x)
    for _ in range(3):
        head(feats, targets=mk())
    torch.cuda.synchronize()
    with Count() as c:
        head(feats, targets=mk())
    torch.cuda.synchronize()
tot = sum(c.src.values())
print(f"{'first clip' if first else 'prompted clip'}: {tot} ATen operator calls that launch (views and metadata excluded)")
byfn = collections.Counter()
for w, n in c.src.items():
    byfn[w.split(":")[0] + " " + w.split(" ")[-1]] += n
print("-- by function")
for w, n in byfn.most_common(40):
    print(f"{n:5d}  {w}")
print("-- by line")
for w, n in c.src.most_common(90):
    top = ", ".join(f"{k} x{v}" for k, v in c.names[w].most_common(4))
    print(f"{n:5d}  {w[:80]:80s} {top[:120]}")
