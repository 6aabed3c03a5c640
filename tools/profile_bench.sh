#!/bin/bash
# Runs on the GPU box: rocprofv3 passes over bench.py.
#   1. --kernel-trace --stats, one queue (--no-overlap)   -> per-dispatch kernel durations (what roofline `single_queue` quotes)
#   2. --kernel-trace, two queues (bench.py's default)     -> dispatches overlap: start-to-start interval = step time
#   3. --pmc FETCH_SIZE, one queue                         -> HBM read traffic   (own pass: TCC slot budget)
#   4. --pmc WRITE_SIZE, one queue                         -> HBM write traffic  (own pass)
# Output: gpurun_out/prof_$1/{trace1q,trace2q,fetch,write}/...  (tools/summarize_profile.py copies the summaries into profiles/)
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BASE="python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-single-queue-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1q" -- $BASE --no-overlap > "$OUT/trace1q_bench.json" 2> "$OUT/trace1q.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace2q" -- $BASE > "$OUT/trace2q_bench.json" 2> "$OUT/trace2q.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BASE --no-overlap > "$OUT/fetch_bench.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BASE --no-overlap > "$OUT/write_bench.json" 2> "$OUT/write.err"
python3 tools/summarize_profile.py "$TAG" > "$OUT/summary.txt" 2>&1
tail -40 "$OUT/summary.txt"
