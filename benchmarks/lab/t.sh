cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" "FZ_BITS_QCAP=64" "FZ_BITS_QCAP=96"; do
  echo "== $v"
  env $v rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $R/benchmarks/regimes.py --only "54,8;100,20" --mib 256 --reps 3 2>&1 | grep '^{'
  python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('/tmp/tr/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:60], r['LDS_Block_Size'], r['Grid_Size'], r.get('VGPR_Count',''))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v)/1e3, 'us')
PY
  rm -rf /tmp/tr
done
