#!/bin/bash
# GPU-box check: parity tests, then the bench line.  Usage: scripts/gpu_check.sh <tag> [bench args]
tag=${1:-x}; shift
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests_$tag.log 2>&1 || { tail -30 gpurun_out/gputests_$tag.log; exit 1; }
tail -2 gpurun_out/gputests_$tag.log
timeout -k 10 300 python bench.py "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -20 gpurun_out/bench_$tag.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print("ms/step %.3f  value %.4g  device ms %.3f" % (d["ms_per_step"], d["value"], d["device_ms_per_iteration"]))
print({k: round(v, 4) for k, v in d["phases_ms_per_iteration"].items()})
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), "eval frac", round(d["eval"]["roofline"]["frac"], 4))
PY
