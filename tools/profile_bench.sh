#!/bin/bash
# Runs on the GPU box: rocprofv3 passes over bench.py (same command every time).
#   1. --kernel-trace --stats          -> per-kernel durations
#   2. --pmc FETCH_SIZE                -> HBM read traffic   (own pass: TCC slot budget)
#   3. --pmc WRITE_SIZE                -> HBM write traffic  (own pass)
# Output: gpurun_out/prof_$1/{trace,fetch,write}/...  (copy the summaries into profiles/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-two-queues"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch_bench.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write_bench.json" 2> "$OUT/write.err"
ls -R "$OUT" | head -40
