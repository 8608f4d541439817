"""Condense rocprofv3 outputs (kernel stats + PMC passes) into a markdown summary (profiles/<round>_summary.md)."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
which = sys.argv[2] if len(sys.argv) > 2 else "cmdm"
TARGET = {"cmdm": "tools/pmc_target.py (12 native steps, B = 32, one stream: launches of M = 10432 rows)",
          "cdm": "tools/pmc_target.py cdm (2 x 12 native CDM steps, B = 32, N = 8192, one stream)",
          "cdm_h": "tools/pmc_target.py cdm_h (HUMANISE variant)",
          "points": "tools/pmc_target.py points (TransitionDown 32 -> 64, k = 16, B = 32, N = 8192 -> 2048 and -> 1024, three repetitions each: means over both strides)"}.get(which, which)


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")     # non-template kernels carry no "void"
    return n.split("(")[0]


print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
try:
    print("bench line under rocprof: `" + open(os.path.join(out, "bench_under_rocprof.json")).read().strip()[:400] + " ...`\n")
except OSError:
    pass
st = glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv"))
if st:
    print("## kernel-trace --stats (top kernels)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for r in list(csv.DictReader(open(st[0])))[:14]:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(out, tag, "*", "*counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == ctr:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    print(f"\n## {ctr} per launch (rocprofv3 units: KiB; FETCH_SIZE on gfx950 counts 64 B per 128 B request -> x2 for wide streaming reads)\n")
    print("| kernel | launches | mean KiB | mean MB (raw) |\n|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:10]:
        print(f"| `{k}` | {len(v)} | {sum(v)/len(v):.0f} | {sum(v)/len(v)*1024/1e6:.2f} |")
# machine-readable HBM traffic of every GEMM kernel for bench.py's roofline.traffic (it picks the entry of its dominant kernel)
try:
    per = collections.defaultdict(dict)
    for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        fl = glob.glob(os.path.join(out, tag, "*", "*counter_collection.csv"))
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fl[0])):
            if r["Counter_Name"] == ctr:
                agg[short(r["Kernel_Name"]).replace(" ", "")].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            per[k][ctr] = sum(v) / len(v) * 1024
            per[k]["launches"] = len(v)
    kernels = {k: {"read_bytes_per_launch": 2 * v["FETCH_SIZE"], "write_bytes_per_launch": v["WRITE_SIZE"],
                   "bytes_per_launch": 2 * v["FETCH_SIZE"] + v["WRITE_SIZE"], "launches": v["launches"]}
               for k, v in per.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    json.dump({"kernels": kernels, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH x2 (gfx950), " + TARGET},
              open(os.path.join(out, "traffic.json"), "w"), indent=1)
except Exception as e:   # noqa
    print(f"(traffic.json not written: {e})")
f = glob.glob(os.path.join(out, "pmc_sq", "*", "*counter_collection.csv"))
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("\n## SQ counters per launch (means)\n\n| kernel | MFMA busy / (cycles x 1024 SIMDs) | waves/SIMD | WAIT_ANY % | WAIT_INST % | ACTIVE % | LDS conflict |\n|---|---|---|---|---|---|---|")
    for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0])))[:8]:
        m = {a: sum(b) / len(b) for a, b in c.items()}
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        print(f"| `{k}` | {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)/(cyc*1024):.3f} | {wc*4/cyc/1024:.2f} | {100*m.get('SQ_WAIT_ANY',0)/wc:.1f} | "
              f"{100*m.get('SQ_WAIT_INST_ANY',0)/wc:.1f} | {100*m.get('SQ_ACTIVE_INST_ANY',0)/wc:.1f} | {m.get('SQ_LDS_BANK_CONFLICT',0):.0f} |")
