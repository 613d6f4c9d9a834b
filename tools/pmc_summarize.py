"""Average rocprofv3 --pmc counter values per dispatch and kernel (input: directory tree with
*counter_collection.csv files; output: JSON {kernel: {counter: mean per dispatch, ...}})."""
import csv, glob, json, os, re, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
            name = name.replace("(anonymous namespace)::", "").replace("void ", "")
            name = re.sub(r"\(.*$", "", name).strip()
            cn, cv = r.get("Counter_Name"), r.get("Counter_Value")
            if cn is None or cv is None:
                continue
            did = r.get("Dispatch_Id") or r.get("Dispatch_ID")
            acc[name][cn].append((did, float(cv)))
out = {}
for k, d in acc.items():
    if not any(s in k for s in ("coarse_", "knn_", "topk_", "sample_thr", "seq_", "ivf_", "gif_", "lif_", "bank_")):
        continue
    e = {}
    for cn, vals in d.items():
        per = defaultdict(float)            # a counter may be reported per XCD / dimension: sum per dispatch
        for did, v in vals:
            per[did] += v
        e[cn] = sum(per.values()) / max(1, len(per))
        e["dispatches_averaged"] = len(per)
    out[k] = e
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("kernels:", len(out))
for k in sorted(out):
    if "coarse_scan" in k or "filter_v2" in k:
        print(k, {c: round(v, 1) for c, v in out[k].items()})
