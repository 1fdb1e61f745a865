# where do k_fast_score's wave cycles go: SQ wait / active counters in separate passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/stall; mkdir -p $O
SHORT="--config ${1:-2} --steps 2 --warmup 1 --no-cpu-baseline --no-subrecords --no-verify"
rocprofv3 --list-avail > $O/avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/$O/p$i -- python3 bench.py $SHORT > /dev/null 2> $O/p$i.err
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/stall/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("uvo::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())})
PY
