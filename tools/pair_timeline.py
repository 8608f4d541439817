"""Timeline of the paired launch ("if it loses, the evidence must be a timeline of that launch").

    python tools/pair_timeline.py <kernel_trace.csv of the unpaired two-stream loop> <kernel_trace.csv of the paired loop>

Input: rocprofv3 --kernel-trace CSVs of tools/pmc_target.py (AFM_PROFILE_STREAMS=2, AFM_PROFILE_TILE=5, AFM_PROFILE_PAIR=0 / 1; 12 steps).
Output (markdown): for one steady-state layer of each run, every kernel with its queue, start and end relative to the layer's first kernel;
for the paired run the idle spans the two cross-stream edges leave around `gemm_f32_split_bf16_pair` (last kernel end on EITHER queue ->
pair start; pair end -> first kernel start on either queue), and the per-layer spans of both schedules."""
import csv
import statistics
import sys


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("kernel_name") or ""
            rows.append(dict(name=name, q=r.get("Queue_Id") or r.get("queue_id") or "?", t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r["t0"])
    return rows


def short(n):
    for key, tag in (("split_bf16_pair", "PAIR(out_proj A + linear1 B)"), ("mha_fwd_split", "attention"), ("randn", "randn"), ("expand_schedule", "schedule"),
                     ("prologue", "prologue")):
        if key in n:
            return tag
    if "gemm_f32_split_bf16<" in n:
        a = n[n.index("<") + 1:].split(",")
        return f"gemm {a[0].strip()}x{a[1].strip()}" + (" split-K" if len(a) > 4 and a[4].strip() not in ("1", "1>") else "")
    return n[:40]


def layers(rows):
    """Split the trace at attention launches: a 'layer window' = from one attention start of a queue to the next attention start of the same queue."""
    att = [i for i, r in enumerate(rows) if "mha_fwd_split" in r["name"]]
    return att


def report(tag, rows, out):
    att = layers(rows)
    if len(att) < 40:
        out.append(f"{tag}: too few attention launches ({len(att)})\n")
        return None
    # steady state: take the window between the attention launches 60 % into the trace, spanning one layer of BOTH queues (4 attention launches)
    k = int(len(att) * 0.6)
    i0, i1 = att[k], att[k + 4]
    base = rows[i0]["t0"]
    out.append(f"### {tag}: kernels between two attention launches, two layers of both sub-batches (us relative to the first)\n")
    out.append("| queue | kernel | start | end | duration |\n|---|---|---|---|---|")
    for r in rows[i0:i1]:
        out.append(f"| {r['q']} | {short(r['name'])} | {(r['t0'] - base) / 1e3:.1f} | {(r['t1'] - base) / 1e3:.1f} | {(r['t1'] - r['t0']) / 1e3:.1f} |")
    out.append("")
    # whole-trace statistics
    pairs = [i for i, r in enumerate(rows) if "split_bf16_pair" in r["name"]]
    stats = {}
    if pairs:
        before, after, dur = [], [], []
        for i in pairs:
            prev_end = max(r["t1"] for r in rows[max(0, i - 6):i])
            nxt = [r["t0"] for r in rows[i + 1:i + 4]]
            before.append((rows[i]["t0"] - prev_end) / 1e3)
            if nxt:
                after.append((min(nxt) - rows[i]["t1"]) / 1e3)
            dur.append((rows[i]["t1"] - rows[i]["t0"]) / 1e3)
        stats = dict(pair_launches=len(pairs), pair_us_median=statistics.median(dur), idle_before_pair_us_median=statistics.median(before),
                     idle_after_pair_us_median=statistics.median(after))
    # time per step: attention launches per step = 2 sub-batches x 5 layers
    steps = len(att) // 10
    t_first, t_last = rows[att[10]]["t0"], rows[att[10 * (steps - 1)]]["t0"]
    stats["us_per_step"] = (t_last - t_first) / 1e3 / (steps - 2) if steps > 2 else None
    # busy fraction: union of kernel intervals / span
    span0, span1 = rows[att[10]]["t0"], rows[att[10 * (steps - 1)]]["t0"]
    iv = sorted((max(r["t0"], span0), min(r["t1"], span1)) for r in rows if r["t1"] > span0 and r["t0"] < span1)
    busy, cur0, cur1 = 0, None, None
    for a, b in iv:
        if cur1 is None or a > cur1:
            if cur1 is not None:
                busy += cur1 - cur0
            cur0, cur1 = a, b
        else:
            cur1 = max(cur1, b)
    if cur1 is not None:
        busy += cur1 - cur0
    stats["device_idle_fraction"] = round(1 - busy / (span1 - span0), 4)
    both = 0
    # time with kernels of BOTH queues in flight
    qs = sorted({r["q"] for r in rows if "mha_fwd_split" in r["name"]})
    if len(qs) >= 2:
        ev = []
        for r in rows:
            if r["q"] in qs[:2] and r["t1"] > span0 and r["t0"] < span1:
                ev.append((max(r["t0"], span0), 1, r["q"])); ev.append((min(r["t1"], span1), -1, r["q"]))
        ev.sort()
        cnt = {q: 0 for q in qs[:2]}
        last = span0
        for t, d, q in ev:
            if all(v > 0 for v in cnt.values()):
                both += t - last
            cnt[q] += d
            last = t
        stats["both_queues_busy_fraction"] = round(both / (span1 - span0), 4)
    out.append(f"{tag}: " + ", ".join(f"{k} = {v if not isinstance(v, float) else round(v, 2)}" for k, v in stats.items()) + "\n")
    return stats


def main():
    out = ["# Paired launch (afm_linear_pair): timeline from rocprofv3 --kernel-trace\n"]
    a = report("unpaired two-stream loop, 128x128 tiles", load(sys.argv[1]), out)
    b = report("paired loop (out_proj A + linear1 B in one launch)", load(sys.argv[2]), out)
    print("\n".join(out))


if __name__ == "__main__":
    main()
