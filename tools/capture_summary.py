"""Print the figures the docs quote from the capture files under profiles/ (tools/final_round.sh + tools/install_capture.sh):  python tools/capture_summary.py [tag]"""
import csv, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"

def load(f):
    rows = [r for r in csv.reader(l for l in open(f) if not l.startswith('#'))]
    return [dict(zip(rows[0], r)) for r in rows[1:]]

def family(k):
    return ("rb16" if "conv_rb16" in k else "sk64" if "conv_sk_kernel<2, 2" in k else "rvq" if "rvq_encode" in k else "ou16" if "ou16" in k
            else "oc16" if "oc16" in k else "small")

for name in ("serial", "steady"):
    R = load(f"profiles/{tag}_kernel_stats_{name}.csv")
    steps = 40
    cat = {}
    for r in R:
        c = cat.setdefault(family(r["kernel"]), [0.0, 0.0]); c[0] += float(r["total_us"]) / steps; c[1] += int(r["launches"]) / steps
    print(name, "sum us/step", round(sum(float(r["total_us"]) for r in R) / steps, 1), "launches", sum(int(r["launches"]) for r in R) / steps,
          {k: (round(v[0], 1), v[1]) for k, v in cat.items()})
    sk = [r for r in R if family(r["kernel"]) == "sk64"]
    print("  sk64 mean us", round(sum(float(r["total_us"]) for r in sk) / sum(int(r["launches"]) for r in sk), 2), "stage 0:", [r["avg_us"] for r in sk if r["grid_threads"] == "122880"])
    for r in R:
        if family(r["kernel"]) in ("ou16", "oc16", "small"):
            print("    ", r["kernel"][:56], r["avg_us"], r["launches"])
for l in open(f"profiles/{tag}_pmc_traffic.csv"):
    if "ou16" in l or "oc16" in l:
        print(l.strip()[:150])
for name in ("serial", "steady"):
    print("T5", name, [(r["kernel"][:32], r["avg_us"], r["launches"]) for r in load(f"profiles/{tag}_kernel_stats_T5_{name}.csv") if "up16" in r["kernel"] or "oc16" in r["kernel"]])
for f in (f"profiles/{tag}_bench_latest.json", f"profiles/{tag}_bench_20_steps.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1]); s = d["summary"]
    print(f, d["value"], d["ms_per_step"], {k: s[k] for k in ("unguarded", "guard_direct_calls", "guard_every_step_synchronised", "exact_f32", "latency_ms", "launches_per_step")})
    print("  ", s["north_star_kernel"]); print("  ", {k: s["roofline"][k] for k in ("frac", "frac_serial", "traffic_bytes_per_launch")})
    print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["more_cores"]["value"])
