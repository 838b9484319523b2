"""Human-readable digest of a bench.py JSON line.  usage: python tools/bench_summary.py FILE"""
import json
import sys

try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print("bench FAILED", e)
    sys.exit(0)
print("pairs/s %.0f ms/step %.3f n_gpus %d" % (d["value"], d["ms_per_step"], d["n_gpus"]))
for k in ("timed", "roofline", "step_split_ms", "parity", "strong_scaling", "collective"):
    if k in d:
        print("  ", k, json.dumps(d[k])[:700])
for name, leg in d.get("other_regimes", {}).items():
    t = leg.get("timed", {})
    print("   leg %-14s %9.0f pairs/s %7.3f ms/step  blocks %s spread %.3f settle %s%s" % (
        name, leg["pairs_per_s"], leg["ms_per_step"], t.get("blocks"), t.get("spread_rel") or 0, t.get("settle_block_ms_per_step"),
        "  UNSTABLE" if leg.get("unstable") else ""))
for row in d.get("realistic_series", {}).get("rows", []):
    t = row.get("timed", {})
    print("   realistic B=%-4d %9.0f pairs/s %7.3f ms/step  blocks %s spread %.3f settle %s%s" % (
        row["claims"], row["pairs_per_s"], row["ms_per_step"], t.get("blocks"), t.get("spread_rel") or 0,
        t.get("settle_block_ms_per_step"), "  UNSTABLE" if row.get("unstable") else ""))
for leg in d.get("other_configs", []):
    t = leg.get("timed", {})
    rf = leg.get("roofline", {})
    print("   cfg %-60s %9.0f pairs/s %7.3f ms/step  blocks %s spread %.3f settle %s  %s frac %.3f parity %s%s%s" % (
        leg["workload"][:60], leg["pairs_per_s"], leg["ms_per_step"], t.get("blocks"), t.get("spread_rel") or 0,
        len(t.get("settle_block_ms_per_step", [])), rf.get("kernel"), rf.get("frac") or 0,
        json.dumps({k: leg.get("parity", {}).get(k) for k in ("max_abs_logit_diff_vs_cpu_oracle", "graphs_with_real_node_keep_set_mismatch")}),
        "  UNSTABLE" if leg.get("unstable") else "", "  UNSETTLED" if leg.get("unsettled") else ""))
for k, v in d.get("kernels", {}).items():
    r = v.get("achieved_gbps", v.get("achieved_tflops", 0))
    print("   %-16s %7.4f ms/step  %5.1f launches  %8.1f %s  frac %.3f" % (
        k, v["ms_per_step"], v["launches_per_step"], r, "GB/s" if "achieved_gbps" in v else "TF", v["frac"]))
cb = d.get("cpu_baseline")
print("  cpu", json.dumps(cb)[:300])
