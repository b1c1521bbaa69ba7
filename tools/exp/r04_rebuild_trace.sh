#!/bin/bash
# Timeline of the always_rebuild frame (what runs on the stream besides the frame kernel): kernel + memory-copy trace of a short loop.
set -e
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/rebuild_trace" -- python3 "$REPO/bench.py" --steps 40 --warmup 8 --timed-loop-only --always-rebuild > "$OUT/rebuild_trace.log" 2>&1
cd "$REPO"
find gpurun_out/rebuild_trace -name '*.csv' | xargs ls -la
