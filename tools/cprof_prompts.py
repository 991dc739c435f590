import cProfile, os, pstats, sys, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univs_amd import workloads as cases
dev = torch.device("cuda:0")
swin, head = cases.build_model(dev)
case = dict(cases.CFG2, H=736, W=1280)
x = cases.preprocess(cases.cfg2_frames()).to(dev)
tv0 = cases.targets_with_entities(case, first_frame_idx=1, n_ent=10)[0]
tvd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tv0.items()}
with torch.no_grad():
    feats = swin(x)
    for _ in range(2):
        head(feats, targets=[dict(tvd)])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        head(feats, targets=[dict(tvd)])
    torch.cuda.synchronize()
    pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(90)
for line in st.getvalue().splitlines():
    if "univs_amd" in line or "ncalls" in line:
        print(line[:170])
