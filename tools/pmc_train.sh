cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_train; rm -rf $OUT; mkdir -p $OUT
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- python /root/repo/tools/bench_train.py --steps 4 --cpu-steps 0 > $OUT/log.txt 2>&1
f=$(find $OUT/sq -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:44]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in acc.items():
    g = sum(c["GRBM_GUI_ACTIVE"]); n = len(c["GRBM_GUI_ACTIVE"])
    rows.append((g, k, n, sum(c["SQ_LDS_BANK_CONFLICT"]) / n, sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / max(g / 8 * 1024, 1), g / n / 8))
for g, k, n, conf, busy, cyc in sorted(rows, reverse=True)[:16]:
    print(f"{k:46s} n={n:4d} cycles/launch={cyc:10.0f} mfma_busy={busy:.3f} lds_conflict_cycles/launch={conf:12.0f} ({conf / 256 / max(cyc, 1) * 100:.1f}% of CU time)")
PY
