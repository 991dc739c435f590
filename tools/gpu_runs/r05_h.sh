#!/bin/bash
# round 5: XCD-aware (row range, pass) order of the GEMM workgroups: Linear tests, gemmset with the order off / on (twice), bench A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_h
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "linear or conv or mlp" > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log; tail -3 $O/tests.log
for i in 1 2; do
  timeout 600 python tools/gemmset.py --tag off$i --linear-ablate 5 > $O/gemmset_off$i.txt 2> $O/err.txt; tail -1 $O/gemmset_off$i.txt
  timeout 600 python tools/gemmset.py --tag on$i > $O/gemmset_on$i.txt 2>> $O/err.txt; tail -1 $O/gemmset_on$i.txt
done
python - <<PY
import json
rows = {}
for v in ("off1", "on1", "off2", "on2"):
    for l in open("$O/gemmset_%s.txt" % v):
        tag, js = l.split(" ", 1); d = json.loads(js); rows.setdefault(d["name"], {})[v] = d
for n, r in rows.items():
    if "us" in r["off1"]:
        print(f"{n:28s} x{r['off1'].get('per_clip', 1)} off {r['off1']['us']:7.1f} {r['off2']['us']:7.1f}  on {r['on1']['us']:7.1f} {r['on2']['us']:7.1f}")
PY
