#!/bin/bash
# Runs on the GPU box: BASELINE config 5 (512 x 65536-point FFT + 10x interpolation) through bench.py, plain and under
# rocprofv3 (kernel trace; FETCH_SIZE / WRITE_SIZE in separate passes).  tools/c5_summary.py -> profiles/<tag>_c5.json
set -u
TAG=${1:-r02}
OUT=gpurun_out/c5_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --workload c5 --steps 40 --warmup 5 > "$OUT/bench_one_call.json" 2> "$OUT/bench.err"
python3 bench.py --workload c5 --c5-unfused --steps 40 --warmup 5 > "$OUT/bench_two_calls.json" 2>> "$OUT/bench.err"
CMD="python3 bench.py --workload c5 --c5-unfused --steps 20 --warmup 3 --settle-ms 5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.json" 2> "$OUT/write.err"
python3 tools/c5_summary.py "$TAG"
