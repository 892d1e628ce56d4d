#!/bin/bash
# Per-configuration profiles of bench.py on the GPU box (one MI355X): for every single-GPU configuration of the
# headline benchmark a plain run (JSON line), a rocprofv3 kernel trace (reduced to the timed steps by
# tools/prof_summary.py) and, for the training step and the ViTPose-H inference, the two HBM-traffic counter passes
# (merged into one pmc_traffic.json keyed by workload).
# usage: tools/collect_profiles.sh <out-dir> <tag>      (run through gpurun; every GPU step has its own time limit)
set -e
OUT=${1:-gpurun_out/prof}; TAG=${2:-r03}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {   # name, steps, warmup, bench args...
    local name=$1 steps=$2 warm=$3; shift 3
    timeout -k 10 240 python3 bench.py --steps "$steps" --warmup "$warm" "$@" > "$OUT/${TAG}_${name}_bench.json" 2> "$OUT/${name}.err"
    timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$name" -o t -- python3 bench.py --steps "$steps" --warmup "$warm" --no-cpu-baseline --no-default-iter "$@" > "$OUT/trace_$name.log" 2>&1
    python3 tools/prof_summary.py "$OUT/trace_$name/t_results.db" --steps "$steps" --csv "$OUT/${TAG}_${name}_kernel_stats.csv" > "$OUT/${TAG}_${name}_summary.txt"
    rm -rf "$OUT/trace_$name"
    echo "== $name"; head -3 "$OUT/${TAG}_${name}_summary.txt"
}
pmc() {   # name, workload key, bench args...
    local name=$1 wl=$2; shift 2
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${name}_$c" -o pmc -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-default-iter "$@" > "$OUT/pmc_${name}_$c.log" 2>&1
    done
    local gf
    gf=$(python3 -c "import json,sys; d=json.loads(open('$OUT/${TAG}_${name}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(r['algorithmic_gflop_per_launch'] if 'grouped' in r['kernel'] else 0)")
    python3 tools/pmc_traffic.py "$OUT/pmc_${name}_FETCH_SIZE/pmc_counter_collection.csv" "$OUT/pmc_${name}_WRITE_SIZE/pmc_counter_collection.csv" "$OUT/${TAG}_pmc_traffic.json" --workload "$wl" --state "$TAG" --grouped-gflop "$gf" --merge
    rm -rf "$OUT/pmc_${name}_FETCH_SIZE" "$OUT/pmc_${name}_WRITE_SIZE"
    echo "== pmc $name done"
}
run train_B 20 5
run infer_B64 30 5 --mode infer --model B --batch 64
run train_L32 10 4 --model L --batch 32
run infer_H128 10 4 --mode infer --model H --batch 128
run infer_B64_exact 20 5 --mode infer --model B --batch 64 --exact
run infer_B64_fold 30 5 --mode infer --model B --batch 64 --fold-ln
pmc train_B train_B64
pmc infer_H128 infer_H128 --mode infer --model H --batch 128
pmc infer_B64 infer_B64 --mode infer --model B --batch 64
