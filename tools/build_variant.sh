#!/bin/bash
# build_variant.sh <name> [extra hipcc flags...]  -> raytracer-rs_amd/lib/variants/lib_<name>.so (A/B timing builds)
set -e
cd "$(dirname "$0")/../raytracer-rs_amd/csrc"
n=$1; shift
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -c render.hip -o ../lib/variants/render_$n.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/lib_$n.so ../lib/variants/render_$n.o ../lib/grid.o ../lib/frontend.o ../lib/cli.o
rm ../lib/variants/render_$n.o
