#!/bin/bash
# usage: tools/ab_workloads.sh "<flags of build B>"   -- A/B (A = default build) on the c5, c4 and c2 workloads
set -e
for w in c5 c4 c2; do
  n=10000; [ $w = c5 ] && n=40000; [ $w = c4 ] && n=4000
  echo "== $w (A = default build, B = $1)"
  bash tools/ab_build_run.sh "$1" $w $n
done
