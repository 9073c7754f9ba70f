# AO parity tests, then the two occlusion workloads of bench.py (short runs, no CPU baseline).
set -e
O=gpurun_out/ao
mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "ao or occlusion or config5" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
fi
timeout -k 10 120 python3 bench.py --workload bunny_1080p_ao64_closeup --no-extra-workloads --no-cpu-baseline --steps 4 --warmup 1 --frames-per-step 16 > $O/closeup.json
echo closeup done
timeout -k 10 120 python3 bench.py --workload bunny_1080p_ao64 --no-extra-workloads --no-cpu-baseline --steps 20 --warmup 5 > $O/ao64.json
python3 - <<'PY'
import json
for n in ("closeup","ao64"):
    d=json.load(open("gpurun_out/ao/%s.json"%n)); print(n, "us/frame", d["config"]["us_per_frame"], "value", d["value"], "alone", (d["roofline"].get("alone") or {}).get("kernel_ms"))
PY
