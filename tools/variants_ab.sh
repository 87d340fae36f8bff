#!/bin/bash
# Quick bench (tools/bench_quick.sh) of every library in tools/variants/.
cd $GRAFT_REPO_ROOT
for f in tools/variants/lib_*.so; do
echo "== $f"
SGZ_LIB_PATH=$GRAFT_REPO_ROOT/$f bash tools/bench_quick.sh || exit 1
done
