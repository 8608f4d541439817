#!/bin/bash
# LDS / issue counters of the 64x64 nine-product GEMM (is the LDS pipe co-critical?) + kernel traces at B = 4 / 1 (launch gaps)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out/r3j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9a-z]*" | sort -u > $O/avail_sq.txt
for b in 4 1; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_b$b -- python $ROOT/bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 > $O/trace_b$b.log 2>&1
  python - $O/trace_b$b <<'PY'
import csv,glob,sys,collections
fs=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)
rows=[]
for f in fs: rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//10:4*len(rows)//10]            # inside the timed region (the profiled roofline pass comes last)
gaps=[];dur=collections.defaultdict(list)
for a,b in zip(rows,rows[1:]):
    gaps.append(int(b['Start_Timestamp'])-int(a['End_Timestamp']))
    dur[a['Kernel_Name'][:60]].append(int(a['End_Timestamp'])-int(a['Start_Timestamp']))
gaps.sort()
n=len(gaps)
print(sys.argv[1].split('/')[-1],'kernels',n,'gap ns p10/p50/p90/mean',gaps[n//10],gaps[n//2],gaps[9*n//10],sum(gaps)//n,'span_us',(int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1e3)
for k,v in sorted(dur.items(),key=lambda kv:-sum(kv[1])): print('   %-62s n=%5d mean %.1f us total %.2f ms'%(k,len(v),sum(v)/len(v)/1e3,sum(v)/1e6))
PY
  find $O/trace_b$b -name "*.csv" -size +2M -delete
done

timeout 200 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $O/hip_b4 -- python $ROOT/bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 > $O/hip_b4.log 2>&1
f=$(find $O/hip_b4 -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
find $O/hip_b4 -name "*.csv" -size +2M -delete
