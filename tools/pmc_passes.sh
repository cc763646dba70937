#!/bin/bash
# PMC counter passes (one rocprofv3 --pmc run per counter group; never combined with tracing) over a small python tool.
# usage: bash tools/pmc_passes.sh <tag> <tool.py> ; env passes through.  Prints per-kernel averages of every counter.
tag=$1; tool=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_I8"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum TCC_TAG_STALL_sum"
 "TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_BUSY_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT TCC_EA_RD_UNCACHED_32B_sum TCC_EA_RDREQ_DRAM_sum TCC_EA_WRREQ_DRAM_sum"
)
i=0
for g in "${groups[@]}"; do
  if [ -n "$PMC_GROUPS" ] && [[ " $PMC_GROUPS " != *" $i "* ]]; then i=$((i+1)); continue; fi      # PMC_GROUPS="0 3": only those counter groups
  d=$out/g$i; mkdir -p $d
  timeout -k 10 150 rocprofv3 --pmc $g --output-format csv -d $d -- python3 $root/$tool > $d/log.txt 2>&1 || echo "group $i failed: $(tail -2 $d/log.txt | tr '\n' ' ')"
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, pathlib, sys, collections
root = pathlib.Path(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in root.rglob('*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-60:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if not any(t in k for t in ('conv', 'attention', 'sla', 'tail', 'attn', 'wgrad', 'norm')): continue
    print('==', k)
    for c, v in sorted(d.items()):
        print(f'   {c:40s} n={len(v):3d} avg={sum(v)/len(v):16.1f}')
PY
