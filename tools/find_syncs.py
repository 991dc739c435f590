"""List the implicit host synchronisations of one prompted clip (torch.cuda.set_sync_debug_mode)."""
import collections, os, sys, traceback, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univs_amd import workloads as cases
dev = torch.device("cuda:0")
swin, head = cases.build_model(dev)
case = dict(cases.CFG2, H=736, W=1280)
x = cases.preprocess(cases.cfg2_frames()).to(dev)
tv0 = cases.targets_with_entities(case, first_frame_idx=1, n_ent=10)[0]
tg = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tv0.items()}]
with torch.no_grad():
    feats = swin(x)
    head(feats, targets=[dict(tg[0])])
    torch.cuda.synchronize()
    sites = collections.Counter()
    def hook(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" in str(message):
            st = [f for f in traceback.extract_stack() if "univs_amd" in f.filename]
            st = [f for f in st if "find_syncs" not in f.filename]
            sites[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:])] += 1
    warnings.showwarning = hook
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    head(feats, targets=[dict(tg[0])])
    torch.cuda.set_sync_debug_mode("default")
for k, v in sites.most_common(40):
    print(v, k)
