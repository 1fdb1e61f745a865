# one PMC counter set for one kernel of the default bench command:  bash tools/pmc_one.sh "WRITE_SIZE FETCH_SIZE" k_fast_score [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C="$1"; K="$2"; shift 2
rm -rf /tmp/pw; timeout 240 rocprofv3 --pmc $C --output-format csv -d /tmp/pw -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-subrecords --no-verify "$@" > /dev/null 2>&1
python3 - "$K" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/pw/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v)/len(v),1) for k,v in acc.items()})
PY
