#!/bin/bash
# Builds instrumented copies of the library next to this script (the product library is not touched):
#   libvitpose_clk.so   -DVP_CLOCK_PROBE=1   core clock of the GEMM K loops          -> clk_summary.py
#   libvitpose_clk2.so  -DVP_CLOCK_PROBE=2   per-workgroup timeline of a launch      -> tile_timeline.py, attn_timeline.py
#   libvitpose_clk3.so  the same + -DVP_ABLATE_SAMETILE (every tile reads the operand rows of tile 0: operands L2-resident)
# Use one with VITPOSE_HIP_LIB=<path> in front of tools/gemm_bench.py, bench.py or tools/probes/tile_timeline.py.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../vitpose_amd/csrc"
make -C "$SRC" > /dev/null
TMP="$(mktemp -d)"
OTHERS=$(ls "$SRC"/build/*.o | grep -v '/gemm.o$' | grep -v '/attention.o$')
build() {  # name, flags: the GEMM and attention sources carry the probe hooks
    for f in gemm attention; do
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$SRC" -I"$SRC/../../include" -Wno-unused-result -Wno-unused-variable \
            $2 -c "$SRC/$f.hip" -o "$TMP/$1.$f.o"
    done
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/$1.so" $OTHERS "$TMP/$1.gemm.o" "$TMP/$1.attention.o"
}
build libvitpose_clk "-DVP_CLOCK_PROBE=1" &
build libvitpose_clk2 "-DVP_CLOCK_PROBE=2" &
build libvitpose_clk3 "-DVP_CLOCK_PROBE=2 -DVP_ABLATE_SAMETILE" &
wait
rm -rf "$TMP"
ls -la "$HERE"/*.so
