#!/bin/bash
# verification run: whole GPU suite, smoke, default bench (with the side configs)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/verify
python -m pytest tests -m gpu -q > gpurun_out/verify/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/verify/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/verify/pytest.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/verify/smoke.log 2>&1; tail -2 gpurun_out/verify/smoke.log
timeout 1200 python bench.py > gpurun_out/verify/bench_default.json 2> gpurun_out/verify/bench_default.err; head -c 400 gpurun_out/verify/bench_default.json; echo
python - <<'PY'
import json
d = json.load(open("gpurun_out/verify/bench_default.json"))
print(json.dumps(d.get("configs"))[:600]); print(json.dumps(d.get("cpu_baseline"))[:400]); print(json.dumps(d.get("parity"))[:300])
PY
