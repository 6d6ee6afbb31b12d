#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into text: per-kernel dispatch statistics
(--kernel-trace --stats) and per-kernel PMC counter averages (--pmc ...).  Dispatches are grouped by (kernel, launch grid):
the same kernel serves 512-point input generation and the 2^20-point leg, and an average over both describes neither.
usage: rocpd_summary.py out.txt db1 [db2 ...]      (environment BLSMI_COMMIT: recorded in the counters file)"""
import sqlite3, sys, collections

def kernels(db):
    q = """select s.kernel_name, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), sum(d.end - d.start),
                  max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.private_segment_size), d.grid_size_x, max(d.workgroup_size_x)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, d.grid_size_x order by 6 desc"""
    return list(db.execute(q))

def pmcs(db):
    q = """select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value), d.grid_size_x, count(distinct d.id)
           from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name, d.grid_size_x, p.name order by 1, 6, 2"""
    return list(db.execute(q))

import json, re
counters = {}            # kernel (demangled short name) -> {field: value}: the machine-readable twin of the text summary (bench.py reads it)
def short(name):
    m = re.match(r"_Z(\d+)", name)                        # Itanium mangling: _Z<length><identifier><parameters>
    return name[m.end():m.end() + int(m.group(1))] if m else name.split("(")[0]

def rec(kernel, grid):
    """the record of (kernel, grid): counters[kernel]["by_grid"][grid]; the top level of counters[kernel] mirrors the grid that took most time"""
    return counters.setdefault(short(kernel), {"by_grid": {}})["by_grid"].setdefault(str(grid), {"grid": grid})

out = open(sys.argv[1], "w")
for path in sys.argv[2:]:
    db = sqlite3.connect(path)
    out.write("== %s\n" % path)
    ks = kernels(db)
    tot = sum(k[5] for k in ks) or 1
    out.write("%-52s %6s %12s %12s %12s %7s %5s %5s %5s %8s %8s\n" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "agpr", "sgpr", "scratchB", "grid"))
    for k in ks:
        if "kt" in path.split("/")[-1] or "prof_" in path:
            rec(k[0], k[10]).update(calls=k[1], avg_ns=k[2], min_ns=k[3], total_ns=k[5], vgpr_rocprof=k[6], agpr=k[7], sgpr=k[8], scratch_bytes_per_lane=k[9])
        if k[5] / tot < 0.002: continue
        out.write("%-52s %6d %12.0f %12d %12d %6.2f%% %5d %5d %5d %8d %8d\n" % (k[0][:52], k[1], k[2], k[3], k[4], 100.0 * k[5] / tot, k[6], k[7], k[8], k[9], k[10]))
    pm = pmcs(db)
    if pm:
        out.write("-- PMC (per-dispatch average, summed over instances/XCDs as stored)\n")
        agg = collections.OrderedDict()
        for kn, pn, c, av, sm, grid, ndisp in pm:
            agg.setdefault((kn, grid), []).append((pn, c, av, sm, ndisp))
        for (kn, grid), rows in agg.items():
            n_disp = max(1, max(r[4] for r in rows))
            out.write("%s grid=%d (dispatches=%d)\n" % (kn[:70], grid, n_disp))
            for pn, c, av, sm, _ in rows:
                out.write("    %-28s samples=%-6d sum/dispatch=%.6g\n" % (pn, c, sm / n_disp))
                r = rec(kn, grid); r[pn] = sm / n_disp; r["pmc_grid"] = grid
    out.write("\n")
out.close()
for k, v in counters.items():                                  # top level = the grid that took most time in the kernel trace (else the largest grid)
    best = max(v["by_grid"].values(), key=lambda r: (r.get("total_ns", 0), r["grid"]))
    for f, x in best.items():
        v[f] = x
import hashlib, os
# host-side translation units: no kernel in them, a change there cannot make a counter or a kernel duration stale
HOST_ONLY = ("blsmi.hip", "verify_host.inc", "kernels.h")


def source_digest():
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bls_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cuh", ".inc", ".h", ".py")) and not f.startswith("lat_programs") and f not in HOST_ONLY:
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]
jpath = sys.argv[1].rsplit(".", 1)[0].replace("_rocprof_summary", "") + "_counters.json"
json.dump({"note": "per-kernel averages per dispatch from rocprofv3 (kernel trace + separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them), "
                   "one record per (kernel, launch grid) under by_grid, the top level mirroring the grid that took most time; written by tools/rocpd_summary.py",
           "commit": os.environ.get("BLSMI_COMMIT"), "source_digest": source_digest(),
           "kernels": {k: v for k, v in counters.items() if k.startswith("k_")}}, open(jpath, "w"), indent=1, sort_keys=True)
print(open(sys.argv[1]).read())
print("wrote", jpath)
