# AddressSanitizer pass over the HOST side of the library (ordering, symbolic analysis, KKT assembly: the g++-compiled
# objects): builds them with -fsanitize=address next to the product's HIP objects in a scratch copy of the package and
# runs the CPU test suite (and, optionally, a host-only setup of a large instance) against it.  No GPU needed.
# usage: bash tools/asan_host.sh [scratch dir]      (the product library must have been built: __graft_entry__.build())
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/chip_asan}
rm -rf "$W" && mkdir -p "$W/pkg"
cp -r "$REPO/clarabel.rs_amd" "$REPO/tests" "$REPO/oracle" "$REPO/include" "$REPO"/*.md "$REPO/__graft_entry__.py" "$REPO/bench.py" "$REPO/BASELINE.json" "$W/pkg/"
rm -f "$W/pkg/clarabel.rs_amd/libclarabel_hip.so"
C="$REPO/clarabel.rs_amd/csrc"
for f in amd_order symbolic kkt_assembly switches; do
    g++ -O1 -g -std=c++17 -fPIC -DCHIP_TESTING -fsanitize=address -fno-omit-frame-pointer -c "$C/$f.cpp" -I"$C" -o "$W/$f.o"
done
# (the HIP-compiled objects of the product's TESTING=1 build as they are: kernels, engine, C ABI)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address -o "$W/pkg/clarabel.rs_amd/libclarabel_hip.so" \
    "$W/amd_order.o" "$W/symbolic.o" "$W/kkt_assembly.o" "$W/switches.o" "$C/algebra.o" "$C/bundle_factor.o" "$C/bundle_solve.o" "$C/bundle_ir.o" \
    "$C/bundle_gstep.o" "$C/snode.o" "$C/snode_g.o" "$C/cones.o" "$C/engine.o" "$C/capi.o" "$C/kktsystem.o" "$C/comm.o" \
    -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cd "$W/pkg"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests -q -m "not gpu"
