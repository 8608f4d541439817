#!/bin/bash
# cache counters of one x9 GEMM launch form (which level bounds the 64x64 kernel?)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$PWD
mkdir -p gpurun_out/pmc9
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*" | sort -u > $ROOT/gpurun_out/pmc9/avail.txt
wc -l $ROOT/gpurun_out/pmc9/avail.txt
i=0
for cfg in "10432 512 512 3 0" "10432 1536 512 5 0" "1304 512 512 0 0"; do
  i=$((i+1))
  for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TA_BUSY_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout 60 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/pmc9/c${i}_$tag -- $ROOT/tools/kernel_sweep one $cfg 3 > $ROOT/gpurun_out/pmc9/c${i}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv,glob,collections,os
root=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/pmc9'
for d in sorted(glob.glob(root+'/c*_*')):
    if not os.path.isdir(d): continue
    files=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    agg=collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if 'gemm_f32_split' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(d), {k: round(sum(v)/len(v)) for k,v in agg.items()}, open(d+'.log').read().strip().split('\n')[-1][:90] if not agg else '')
PY
find $ROOT/gpurun_out/pmc9 -name "*.csv" -size +2M -delete
