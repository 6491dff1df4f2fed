#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
R=$(pwd)
cd /tmp && DIB_ST_ATTENTION=flash timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_flash -o kt -- python $R/tools/set_transformer_bench.py --batch 2 --particles 4096 --steps 3 > $R/$O/prof_flash.log 2>&1
cd $R; find $O/prof_flash -mindepth 2 -type f -exec mv {} $O/prof_flash/ \;
head -n 14 $O/prof_flash/kt_kernel_stats.csv | cut -c1-200
cd /tmp && DIB_ST_ATTENTION=gemm timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gemm -o kt -- python $R/tools/set_transformer_bench.py --batch 2 --particles 4096 --steps 3 > $R/$O/prof_gemm.log 2>&1
cd $R; find $O/prof_gemm -mindepth 2 -type f -exec mv {} $O/prof_gemm/ \;
head -n 16 $O/prof_gemm/kt_kernel_stats.csv | cut -c1-200
cd /tmp && DIB_ST_ATTENTION=flash timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/prof_flash_s -o s -- python $R/tools/set_transformer_bench.py --batch 2 --particles 4096 --steps 2 > $R/$O/prof_flash_s.log 2>&1
cd $R; find $O/prof_flash_s -mindepth 2 -type f -exec mv {} $O/prof_flash_s/ \;
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/r02h/prof_flash_s/s_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.defaultdict(float); seen=set()
for r in rows:
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if "attn" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); cnt[k]+=1; dur[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
for k,v in agg.items():
    n=cnt[k]; w=v["SQ_WAVE_CYCLES"]
    print(k, "us=%.1f"%(dur[k]/n), "MFMA_busy=%.2f"%(v["SQ_VALU_MFMA_BUSY_CYCLES"]/max(v["GRBM_GUI_ACTIVE"],1)*8/1024), "clk=%.2f"%(v["GRBM_GUI_ACTIVE"]/n/8/(dur[k]/n)/1e3),
          "WAIT_ANY=%.2f"%(v["SQ_WAIT_ANY"]/w), "WAIT_INST=%.2f"%(v["SQ_WAIT_INST_ANY"]/w), "ACTIVE=%.2f"%(v["SQ_ACTIVE_INST_ANY"]/w), "LDSconf=%.2f"%(v["SQ_LDS_BANK_CONFLICT"]/max(v["SQ_ACTIVE_INST_LDS"],1)))
PY
