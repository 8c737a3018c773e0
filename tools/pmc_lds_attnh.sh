#!/bin/bash
# LDS / wait counters of the fp16 attention kernel at the config-5 shape: rocprofv3 --pmc passes over tools/attnh_one.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-q}
export GDX_ATTNH_WAVES=$V
mkdir -p $R/gpurun_out/s6
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/s6/pmc_${V}_$tag -o p --output-format csv -- python3 $R/tools/attnh_one.py 128 521 4 1024 > $R/gpurun_out/s6/pmc_${V}_$tag.log 2>&1 || tail -5 $R/gpurun_out/s6/pmc_${V}_$tag.log
done
cd $R
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/s6/pmc_${V}_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open(f)):
        if "attentionh8" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
    print(f, "avg kernel us %.1f" % (sum(dur)/max(1,len(dur))/1e3))
    for k,v in acc.items(): print(f"   {k:32s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
