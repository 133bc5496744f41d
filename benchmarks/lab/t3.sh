cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FUZZYSEARCH_HIP_LIB=$R/benchmarks/lab/libfzhip_nocols.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d /tmp/pm -- python $R/benchmarks/regimes.py --only "54,8" --reps 20 2>&1 | grep '^{'
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pm/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'fz_' in k:
        print(k, {c: round(sum(v)/len(v)/1e6,2) for c,v in d.items()}, 'M')
PY
