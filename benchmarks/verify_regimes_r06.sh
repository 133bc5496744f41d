#!/bin/bash
# profiles/r06_verify_regimes.txt: the verification regimes with round 5's forms (FZ_NO_BITS=1) and with the bit-vector form —
# regimes.py lines, kernel trace and three PMC passes for DNA m = 54, k = 8 and m = 20, k = 3 / 4, both ways.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/verify_regimes_r06.txt
{
echo "# benchmarks/verify_regimes_r06.sh — 1 GiB DNA (configs[1]'s sequence), Levenshtein n-gram search, synchronous C-ABI calls"
echo "## all six regimes, round 5's verification forms (FZ_NO_BITS=1: register band k <= 4, lane-per-cell k >= 5)"
FZ_NO_BITS=1 python $ROOT/benchmarks/regimes.py --reps 10 2>&1 | grep '^{'
echo "## all six regimes, round 6 (bit-vector columns from k = 3 on and wherever the pattern lets expect dense candidates)"
python $ROOT/benchmarks/regimes.py --reps 20 2>&1 | grep '^{'
echo "## kernel trace + PMC, OLD forms: m = 54 k = 8 (lane-per-cell, 32 lanes per candidate, two launches), m = 20 k = 3 / 4 (register band)"
FZ_NO_BITS=1 bash $ROOT/benchmarks/pmc_regimes.sh r06_vr_old "54,8;20,3;20,4" 1024
echo "## kernel trace + PMC, NEW form: the same searches (fz_scan_kernel<..., 1>)"
bash $ROOT/benchmarks/pmc_regimes.sh r06_vr_new "54,8;20,3;20,4" 1024
} > $OUT 2>&1
tail -5 $OUT
