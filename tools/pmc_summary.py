#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (CSV output).

usage: pmc_summary.py <dir with *FETCH_SIZE*/ and *WRITE_SIZE*/ pass directories> <out.csv>
With `bench.py --pmc-markers` only the launches of the timed steps are counted (see load()).
FETCH_SIZE is doubled (MI355X_MICROARCH.md: on gfx950 it reports half the bytes of 16 B/lane streaming reads); values are KB per launch,
averaged per (kernel, grid size)."""
import collections, csv, glob, os, sys

MARKERS = [0, 0]        # passes in which the two bench.py --pmc-markers launches were found / passes read


def load(d, counter):
    """(kernel, grid) -> [sum, launches] over the dispatches BETWEEN the two marker launches of `bench.py --pmc-markers` (the
    largest-grid `arange` kernel, exactly twice) when they are there, else over all dispatches of the pass."""
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        lo, hi = -1, 1 << 62
        ar = [r for r in rows if "arange" in r["Kernel_Name"]]
        MARKERS[1] += 1
        if ar:
            g = max(int(r["Grid_Size"]) for r in ar)
            ids = sorted(int(r["Dispatch_Id"]) for r in ar if int(r["Grid_Size"]) == g)
            if len(ids) == 2 and g >= 1 << 16:
                lo, hi = ids
                MARKERS[0] += 1
        for r in rows:
            if not lo < int(r["Dispatch_Id"]) < hi:
                continue
            k = (r["Kernel_Name"], int(r["Grid_Size"]))
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc

def source_digest():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as g
    return g.kernel_source_digest()[:16]


def main():
    root, out = sys.argv[1], sys.argv[2]
    fd = [d for d in glob.glob(os.path.join(root, "*FETCH_SIZE*")) if os.path.isdir(d)]
    wd = [d for d in glob.glob(os.path.join(root, "*WRITE_SIZE*")) if os.path.isdir(d)]
    f = load(fd[0], "FETCH_SIZE") if fd else {}
    w = load(wd[0], "WRITE_SIZE") if wd else {}
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 1])[0])):
        if not k[0].startswith("void adk::") and "adk" not in k[0]:
            continue
        fa = f.get(k, [0.0, 0]); wa = w.get(k, [0.0, 0])
        fk = fa[0] / fa[1] if fa[1] else 0.0
        wk = wa[0] / wa[1] if wa[1] else 0.0
        name = k[0].replace("void adk::", "").split("(adk::")[0]
        rows.append((name, k[1], max(fa[1], wa[1]), round(fk), round(2 * fk), round(wk)))
    with open(out, "w") as fh:
        fh.write("# source_digest: %s   (sha256 of audiodec_amd/csrc/*.hip + headers at capture: bench.py marks roofline.traffic stale when the build differs)\n" % source_digest())
        import __graft_entry__ as g_
        fh.write("# schedule_digest: %s   (sha256 of the host sources that decide the launches of a step: pipeline.py, program.py, stream_generator.py, arch.py)\n" % g_.schedule_digest()[:16])
        fh.write("# bench_config: %s\n" % os.environ.get("ADK_PROFILE_CONFIG", "unknown"))
        fh.write("# region: %s\n" % ("launches between the two bench.py --pmc-markers (the timed steps)" if MARKERS[0] == MARKERS[1] and MARKERS[1]
                                     else "ALL launches of the passes (markers not found in every pass)"))
        fh.write("kernel,grid_threads,launches,FETCH_SIZE_KB_avg,FETCH_KB_x2_corrected,WRITE_SIZE_KB_avg\n")
        for r in rows:
            fh.write('"%s",%d,%d,%d,%d,%d\n' % r)
    print(open(out).read())

if __name__ == "__main__":
    main()
