#!/bin/bash
# The two fuzzers over the library as built, with and without the forced kernel variants and builds; one summary line
# each in gpurun_out/fuzz_round.txt.  Each step under a time limit of its own; a failure ends the script.
# FUZZ_SEED_OFFSET=n adds n to every seed (other scenes and cameras than the last run's).
cd "$(dirname "$0")/.."
out=gpurun_out/fuzz_round.txt; : > $out
S=${FUZZ_SEED_OFFSET:-0}
step() { # label, environment assignments..., -- command
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  if env "${envs[@]}" timeout -k 10 400 "$@" > gpurun_out/fuzz_step.log 2>&1; then
    echo "$label: $(tail -1 gpurun_out/fuzz_step.log)" >> $out
  else
    echo "$label: FAILED"; tail -20 gpurun_out/fuzz_step.log; echo "$label: FAILED" >> $out; exit 1
  fi
  tail -1 $out
}
step "parity seed $((31 + S))" -- python3 tools/fuzz_parity.py --frames 1500 --seed $((31 + S))
step "parity seed $((32 + S)), every draw binned" INV_BINS=1 -- python3 tools/fuzz_parity.py --frames 800 --seed $((32 + S))
step "parity seed $((33 + S)), one lane per triangle" INV_SCATTER=1 -- python3 tools/fuzz_parity.py --frames 800 --seed $((33 + S))
step "parity seed $((34 + S)), from the coarse cut" INV_FLAT=0 -- python3 tools/fuzz_parity.py --frames 800 --seed $((34 + S))
step "parity seed $((35 + S)), searching build" INV_BUILD_SEARCH=1 -- python3 tools/fuzz_parity.py --frames 600 --seed $((35 + S))
step "lbvh seed $((41 + S))" -- python3 tools/fuzz_lbvh.py --scenes 300 --seed $((41 + S))
step "lbvh seed $((42 + S)), up to 400 000 triangles" -- python3 tools/fuzz_lbvh.py --scenes 120 --seed $((42 + S)) --max-triangles 400000
step "lbvh seed $((43 + S)), fallback forced" INV_BUILD_TOP_CAP=6 -- python3 tools/fuzz_lbvh.py --scenes 150 --seed $((43 + S))
step "lbvh seed $((44 + S)), searching build" INV_BUILD_SEARCH=1 -- python3 tools/fuzz_lbvh.py --scenes 150 --seed $((44 + S))
echo done
