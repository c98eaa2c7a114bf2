#!/usr/bin/env python3
"""Per-kernel durations of the TIMED STEPS of a `bench.py --pmc-markers` run from a rocprofv3 --kernel-trace CSV.

usage: trace_summary.py <dir or *_kernel_trace.csv> <out.csv> [steps]

bench.py --pmc-markers launches a recognisable ATen kernel (arange over PMC_MARKER_N elements = the largest-grid `arange` of the run,
exactly twice) right before and right after the timed steps; only the dispatches between the two are kept -- model loading, warm-up,
the per-op profile, the single-stream leg and the other-precision leg are not mixed in.  Per kernel (and grid size): launches,
launches per step (when `steps` is given), average / min / max / total duration in us, share of the summed kernel time, registers,
LDS.  Durations are End_Timestamp - Start_Timestamp of the dispatch record (ns): the kernel alone, no launch gap, no event record.
"""
import collections
import csv
import glob
import os
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no *kernel_trace.csv under {src}"
    rows = list(csv.DictReader(open(files[0])))
    ar = [r for r in rows if "arange" in r["Kernel_Name"]]
    lo, hi, marked = -1, 1 << 62, False
    if ar:
        g = max(int(r["Grid_Size_X"]) for r in ar)
        ids = sorted(int(r["Dispatch_Id"]) for r in ar if int(r["Grid_Size_X"]) == g)
        if len(ids) == 2 and g >= 1 << 16:
            lo, hi, marked = ids[0], ids[1], True
    acc = collections.OrderedDict()
    t_first, t_last = None, None
    for r in rows:
        if not lo < int(r["Dispatch_Id"]) < hi:
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        t_first = s if t_first is None else min(t_first, s)
        t_last = e if t_last is None else max(t_last, e)
        name = r["Kernel_Name"].replace("void adk::", "").replace("(anonymous namespace)::", "")
        name = name.split("(adk::")[0].split("(float")[0].split("(int")[0]
        k = (name, int(r["Grid_Size_X"]))
        a = acc.setdefault(k, dict(n=0, tot=0, mn=1 << 62, mx=0, vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]),
                                   lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]), wg=int(r["Workgroup_Size_X"])))
        d = e - s
        a["n"] += 1; a["tot"] += d; a["mn"] = min(a["mn"], d); a["mx"] = max(a["mx"], d)
    total = sum(a["tot"] for a in acc.values()) or 1
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    with open(out, "w") as fh:
        fh.write("# source_digest: %s   (sha256 of audiodec_amd/csrc/*.hip + headers at capture)\n" % g.kernel_source_digest()[:16])
        fh.write("# schedule_digest: %s   (sha256 of the host sources that decide the launches of a step: pipeline.py, program.py, stream_generator.py, arch.py)\n" % g.schedule_digest()[:16])
        fh.write("# bench_config: %s\n" % os.environ.get("ADK_PROFILE_CONFIG", "unknown"))
        fh.write("# region: %s\n" % ("dispatches between the two bench.py --pmc-markers (the timed steps)" if marked else "ALL dispatches (markers not found)"))
        if t_first is not None:
            fh.write("# wall span of the region: %.1f us; summed kernel time %.1f us (kernels of three HIP streams overlap)\n" % ((t_last - t_first) / 1e3, total / 1e3))
        fh.write("kernel,grid_threads,workgroup,launches,launches_per_step,avg_us,min_us,max_us,total_us,share_pct,vgpr,agpr,sgpr,lds_bytes,scratch_bytes\n")
        for (name, grid), a in sorted(acc.items(), key=lambda kv: -kv[1]["tot"]):
            fh.write('"%s",%d,%d,%d,%s,%.2f,%.2f,%.2f,%.1f,%.2f,%d,%d,%d,%d,%d\n' % (
                name, grid, a["wg"], a["n"], ("%.2f" % (a["n"] / steps)) if steps else "", a["tot"] / a["n"] / 1e3, a["mn"] / 1e3, a["mx"] / 1e3,
                a["tot"] / 1e3, 100.0 * a["tot"] / total, a["vgpr"], a["agpr"], a["sgpr"], a["lds"], a["scratch"]))
    print(open(out).read())


if __name__ == "__main__":
    main()
