#!/bin/bash
# Runs on the GPU box: HBM traffic of every kernel of an arbitrary python command.
#   tools/pmc_run.sh <tag> <script.py> [args...]
# Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: TCC slot budget), then a per-kernel table.
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$@" > "$OUT/fetch.out" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$@" > "$OUT/write.out" 2> "$OUT/write.err"
python3 tools/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
