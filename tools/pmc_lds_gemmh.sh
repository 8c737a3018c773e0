#!/bin/bash
# LDS / wait counters of the fp16 256x256 GEMM (kind 0) on the config-5 FFN shape: one rocprofv3 --pmc pass per counter set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export CHECK=0 GDX_GEMMH8T=0 GDX_GEMMH_NOSPLIT=1
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/s5/pmc_$tag -o p --output-format csv -- python3 $R/tools/gemmh_one.py 66688 1024 1024 0 10 > $R/gpurun_out/s5/pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/s5/pmc_$tag.log; }
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/s5/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemmh8b" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for k,v in acc.items(): print(f"   {k:40s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
