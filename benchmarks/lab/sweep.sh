R=$GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python $R/benchmarks/regimes.py --only "54,8;20,4" --reps 20 2>&1 | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['m'],d['k'],'ms',d['ms'],'kernel',d['scan_kernel_ms'])"; }
run A=1
for t in 6 8 16 24; do run FZ_TILES_PER_WG=$t; done
for r in 3 5 6 8; do run FZ_ROUNDS=$r; done
run FZ_TAPER_STEPS=0
run FZ_DUAL_STREAM=1
