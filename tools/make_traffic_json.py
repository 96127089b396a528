"""profiles/<tag>_traffic.json from a tools/profile_round.sh run: `python tools/make_traffic_json.py gpurun_out/<tag> profiles/r02_traffic.json`.
HBM bytes per launch = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request;
the calibration rows of tools/membench in the same file show it); rocprofv3 --kernel-trace --stats average per kernel."""
import csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
src, dst = sys.argv[1], sys.argv[2]
tag = os.path.basename(dst).split("_")[0]
rows = {}
for line in open(os.path.join(src, "pmc_traffic.txt")):
    m = re.match(r"(\S+)\s+(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) mean=\s*([\d.]+)", line)
    if m:
        rows[(m.group(1), m.group(2).strip(), m.group(3))] = float(m.group(5))
stats = {r["Name"]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv")))}
def kern(sub, label):
    f = next(v for (d, k, c), v in rows.items() if d == "enc_fetch" and sub in k and c == "FETCH_SIZE")
    w = next(v for (d, k, c), v in rows.items() if d == "enc_write" and sub in k and c == "WRITE_SIZE")
    avg = next(v for k, v in stats.items() if sub in k)
    return {"fetch_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024), "traffic_bytes": int(2 * f * 1024 + w * 1024), "rocprof_avg_ms": round(avg, 5)}
cal = {k: v for k, v in rows.items() if k[0].startswith("cal_")}
out = {
    "source": f"profiles/{tag}_pmc_traffic.txt + profiles/{tag}_kernel_stats.csv (tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, rocprofv3 --kernel-trace --stats; bench.py --headline-only)",
    "kernel_sources_sha16": bench.kernel_sources_sha16(),
    "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (FETCH_SIZE counts 64 B per 128-B request on gfx950 = half the bytes; calibrated in the same run with tools/membench, rows cal_fetch / cal_write of " + tag + "_pmc_traffic.txt)",
    "workload": "2000-frame 512x512 uint16 synth-v1 stack, per launch",
    "k_encode_fused<uint16_t>": kern("k_encode_fused<unsigned short>", "enc"),
    "k_decode_frames<uint16_t>": kern("k_decode_frames<unsigned short", "dec"),
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
