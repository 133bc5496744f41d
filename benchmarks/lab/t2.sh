cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ONLY=20,4 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pm -- python $R/benchmarks/subs_dense.py 2>&1 | grep subs
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pm/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:70]+' lds='+r['LDS_Block_Size']+' vgpr='+r['VGPR_Count']][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'fz_' in k:
        print(k, {c: round(sum(v)/len(v)/1e6,2) for c,v in d.items()}, 'M')
PY
FZ_BITS_MIN_K=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pm2 -- python $R/benchmarks/regimes.py --only "20,4" --reps 5 2>&1 | grep '^{'
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pm2/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:70]+' lds='+r['LDS_Block_Size']+' vgpr='+r['VGPR_Count']][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'fz_' in k:
        print(k, {c: round(sum(v)/len(v)/1e6,2) for c,v in d.items()}, 'M')
PY
