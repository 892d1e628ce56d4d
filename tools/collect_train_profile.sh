set -e
OUT=gpurun_out/${TAG_DIR:-r2f}; TAG=${TAG:-r02_f}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=train_B; steps=20; warm=5
timeout -k 10 240 python3 bench.py --steps $steps --warmup $warm > "$OUT/${TAG}_${name}_bench.json" 2> "$OUT/${name}.err"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$name" -o t -- python3 bench.py --steps $steps --warmup $warm --no-cpu-baseline > "$OUT/trace_$name.log" 2>&1
python3 tools/prof_summary.py "$OUT/trace_$name/t_results.db" --steps $steps --all $((steps + warm + 1)) --csv "$OUT/${TAG}_${name}_kernel_stats.csv" > "$OUT/${TAG}_${name}_summary.txt"
rm -rf "$OUT/trace_$name"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${name}_$c" -o pmc -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline > "$OUT/pmc_${name}_$c.log" 2>&1
done
python3 tools/pmc_traffic.py "$OUT/pmc_${name}_FETCH_SIZE/pmc_counter_collection.csv" "$OUT/pmc_${name}_WRITE_SIZE/pmc_counter_collection.csv" "$OUT/${TAG}_${name}_pmc_traffic.json" > /dev/null
rm -rf "$OUT/pmc_${name}_FETCH_SIZE" "$OUT/pmc_${name}_WRITE_SIZE"
head -6 "$OUT/${TAG}_${name}_summary.txt"
