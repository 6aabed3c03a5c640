#!/bin/bash
# Runs on the GPU box: BASELINE config 5's transform (512 x 65536-point FFT) under the lab shapes of
# tools/c5_shapes.py -- timing (interleaved A/B), then per shape a kernel trace and FETCH_SIZE / WRITE_SIZE in
# separate rocprofv3 passes -- and the C5 chain through bench.py.  tools/c5_summary.py -> profiles/<tag>_c5.json
set -u
TAG=${1:-r03}
OUT=gpurun_out/c5_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AETH_TUNING=1
timeout -k 10 300 python3 tools/c5_shapes.py > "$OUT/shapes.txt" 2>&1 || exit 1
python3 bench.py --workload c5 --steps 40 --warmup 5 > "$OUT/bench_one_call.json" 2> "$OUT/bench.err" || exit 1
python3 bench.py --workload c5 --c5-unfused --steps 40 --warmup 5 > "$OUT/bench_two_calls.json" 2>> "$OUT/bench.err" || exit 1
i=0
for SHAPE in "default" "groups 64" "parts+groups 64"; do
  D="$OUT/shape$i"; mkdir -p "$D"; echo "$SHAPE" > "$D/name.txt"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 tools/c5_shapes.py --only "$SHAPE" --batch 512 --launches 20 --rounds 1 > "$D/trace.txt" 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/fetch" -- python3 tools/c5_shapes.py --only "$SHAPE" --batch 512 --launches 20 --rounds 1 > "$D/fetch.txt" 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/write" -- python3 tools/c5_shapes.py --only "$SHAPE" --batch 512 --launches 20 --rounds 1 > "$D/write.txt" 2>&1 || exit 1
  i=$((i+1))
done
python3 tools/c5_summary.py "$TAG"
