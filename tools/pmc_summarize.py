"""Average rocprofv3 --pmc counter values per dispatch and kernel.  Input: <dir>/<config>/<pass>/**/
*counter_collection.csv; output: JSON {config: {kernel: {counter: mean per dispatch, ...}}}."""
import csv, glob, json, os, re, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
KEEP = ("coarse_", "knn_", "topk_", "sample_thr", "seq_", "ivf", "gif_", "lif_", "bank_", "kmeans_", "centroid_", "probe_")
out = {}
for cfg in sorted(os.listdir(src)):
    if not os.path.isdir(os.path.join(src, cfg)):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, cfg, "**", "*counter_collection.csv"), recursive=True):
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
    res = {}
    for k, d in acc.items():
        if not any(s in k for s in KEEP):
            continue
        e = {}
        for cn, vals in d.items():
            per = defaultdict(float)            # a counter may be reported per XCD / dimension: sum per dispatch
            for did, v in vals:
                per[did] += v
            e[cn] = sum(per.values()) / max(1, len(per))
            e["dispatches_averaged"] = len(per)
        res[k] = e
    out[cfg] = res
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for cfg, res in out.items():
    for k in sorted(res):
        if "coarse_scan" in k or "filter_v2" in k:
            print(cfg, k, {c: round(v, 1) for c, v in res[k].items()})
