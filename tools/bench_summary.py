"""Short summary of a bench.py JSON line: python tools/bench_summary.py <file>"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "frac", round(r.get("frac", 0), 3),
      "kernel ms", round(r.get("avg_kernel_ms", 0), 4), "issued TF", round(r.get("bf16_tflops_issued", 0) or 0, 1))
s = d.get("secondary") or {}
for k in ("exact_recall_2048q", "exact_recall_256q", "centroid_index_recall_256q", "centroid_index_recall_no_overflow_read",
          "centroid_index_recall_16384q", "rebuild_centroids_ms", "interleaved_store_retrieve_B8",
          "interleaved_store_retrieve_B256", "reference_semantics_write"):
    v = s.get(k)
    if v is not None:
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a not in ("roofline", "note")}
              if isinstance(v, dict) else round(v, 3))
if "neurons" in s:
    print({k: (round(v["ms"], 4), round(v.get("hbm_frac_of_8TBs", 0), 3)) for k, v in s["neurons"].items()})
if "config5_seeding" in s:
    c = s["config5_seeding"]
    print("c5", {k: round(v, 4) if isinstance(v, float) else v for k, v in c["centroid_index"].items()}, "write_s", round(c["write_s"], 4))
if "config2_100k" in s:
    c = s["config2_100k"]
    print("c2", {k: round(v["retrievals_per_s"]) for k, v in c.items() if isinstance(v, dict) and "retrievals_per_s" in v})
if d.get("cpu_baseline"):
    print("cpu", round(d["cpu_baseline"]["value"], 2), d["cpu_baseline"]["gpu_parity_on_sample"])
