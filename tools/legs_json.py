"""profiles/<tag>_legs.json from a tools/profile_legs.sh directory: per leg and route the library's kernels with their rocprofv3
average (--kernel-trace --stats) and their HBM bytes per launch from the TCC counters (separate --pmc passes;
hbm_bytes = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024: FETCH_SIZE counts 64 B per 128-B request on gfx950, see
profiles/README.md), next to the leg's algorithmic bytes and HIP-event time per call.
`python3 tools/legs_json.py gpurun_out/<dir> profiles/r04_legs.json`"""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
src, dst = sys.argv[1], sys.argv[2]
out = {"source": "tools/profile_legs.sh (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes) over tools/leg_prof.py",
       "kernel_sources_sha16": bench.kernel_sources_sha16(),
       "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch (FETCH_SIZE counts 64 B per 128-B request on gfx950; calibration rows: profiles/r04_pmc_traffic.txt)",
       "legs": {}}
def short(n):
    n = n.split("(")[0].replace("void ", "").replace("trpx::", "")
    return n.replace("unsigned short", "u16").replace("unsigned char", "u8").replace("unsigned int", "u32")
for log in sorted(glob.glob(os.path.join(src, "*_*.log"))):
    base = os.path.basename(log)[:-4]
    if base.endswith("_FETCH_SIZE") or base.endswith("_WRITE_SIZE"):
        continue
    m = re.search(r"(\S+) (\S+): ([\d.]+) ms per call, ([\d.]+) of 8 TB/s on (\d+) algorithmic bytes", open(log).read())
    if not m:
        continue
    leg = {"ms_per_call_hip_events": float(m.group(3)), "frac_of_hbm_peak": float(m.group(4)), "algorithmic_bytes": int(m.group(5)), "kernels": {}}
    for r in csv.DictReader(open(os.path.join(src, base + "_kernel_stats.csv"))):
        if "trpx" in r["Name"] and "k_synth" not in r["Name"] and int(r["Calls"]) >= 3:
            leg["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                                "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2)}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(src, f"{base}_{c}.csv")
        if not os.path.exists(f):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if k in leg["kernels"] and len(v) >= 2:
                v = v[1:]                                   # (the first launch also loads the code)
                kib = sum(v) / len(v)
                leg["kernels"][k]["fetch_bytes" if c == "FETCH_SIZE" else "write_bytes"] = int((2 if c == "FETCH_SIZE" else 1) * kib * 1024)
    for k, v in leg["kernels"].items():
        if "fetch_bytes" in v and "write_bytes" in v:
            v["hbm_bytes"] = v["fetch_bytes"] + v["write_bytes"]
    out["legs"][f"{m.group(1)}:{m.group(2)}"] = leg
json.dump(out, open(dst, "w"), indent=1)
for k, v in out["legs"].items():
    ks = ", ".join(f"{n} {d['avg_us']} us" + (f" / {d['hbm_bytes'] / 1e6:.0f} MB" if 'hbm_bytes' in d else "") for n, d in sorted(v["kernels"].items(), key=lambda x: -x[1]["avg_us"])[:5])
    print(f"{k:16s} {v['ms_per_call_hip_events']:.4f} ms ({v['frac_of_hbm_peak']:.3f}) alg {v['algorithmic_bytes'] / 1e6:.0f} MB | {ks}")
