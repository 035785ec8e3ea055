# SQ issue / wait / instruction-fetch counters of one fp32 conv launch (the 64x64-tile class):  bash tools/scratch/pmc_f32.sh n h w cin cout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
G2="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LEVEL_WAVES"
G3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"
i=0
for grp in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcf_$i
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmcf_$i -- python tools/one_conv.py "$@" c3s1 4 > gpurun_out/pmcf_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmcf_$i.log; exit 1; }
  f=$(ls gpurun_out/pmcf_$i/*/*counter_collection.csv | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "igemm" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in rows)
acc = collections.OrderedDict()
for r in rows:
    if int(r["Dispatch_Id"]) == last:
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(rows[-1]["Kernel_Name"][:70])
for k, v in acc.items():
    print(f"  {k:28s} {v:16.0f}")
PY
done
