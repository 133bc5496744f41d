#!/bin/bash
# On the GPU box: time lab builds (ab_scan.py at 1 GiB) -> gpurun_out/lab_<tag>.jsonl
#   benchmarks/lab_run.sh <tag> [ENV=VAL,...:]lib.so ...     (default: every build in benchmarks/lab)
TAG=${1:-lab}; shift
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out
: > gpurun_out/lab_$TAG.jsonl
for SPEC in ${@:-benchmarks/lab/*.so}; do
  L=${SPEC##*:}; ENVS=""
  if [[ "$SPEC" == *:* ]]; then ENVS=$(echo "${SPEC%%:*}" | tr ',' ' '); fi
  echo "# $SPEC" >> gpurun_out/lab_$TAG.jsonl
  env $ENVS FUZZYSEARCH_HIP_LIB=$PWD/$L timeout 300 python benchmarks/ab_scan.py 1024 300 >> gpurun_out/lab_$TAG.jsonl 2>> gpurun_out/lab_$TAG.err
done
cat gpurun_out/lab_$TAG.jsonl
