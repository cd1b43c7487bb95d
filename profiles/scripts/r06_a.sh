#!/bin/bash
# round 6, first GPU call: the FM forward laboratory with the scalar-path prefetch forms (g4s) beside the product's form (g4).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06a
mkdir -p $O
export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/profiles/ubench/fm_fwd_lab
for v in 9 27 28 29 30 31 32 33 34 9; do timeout 60 $LAB $v 40 2>&1 | grep -v "^B "; done > $O/lab.txt
cat $O/lab.txt
