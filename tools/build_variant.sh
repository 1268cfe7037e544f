# Build a variant of libngp_hip.so with extra -D flags for render_fused.hip: bash tools/build_variant.sh <name> [-DFOO=1 ...]
# -> build/var/libngp_<name>.so (select it with NGP_HIP_LIB=...; tools/ab_variants.sh times several)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
V=$ROOT/build/var
mkdir -p $V
cd $ROOT/nerf-navigation_amd/csrc
make -s >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-strict-aliasing -Wno-unused-function \
    -DNGP_BUILD "$@" -c render_fused.hip -o $V/render_fused_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libngp_$name.so ../lib/obj/raymarching.o ../lib/obj/density_grid.o ../lib/obj/gridencoder.o \
    ../lib/obj/shencoder.o ../lib/obj/freqencoder.o ../lib/obj/ffmlp.o ../lib/obj/ffmlp_backward.o ../lib/obj/ffmlp_generic.o ../lib/obj/field_train.o ../lib/obj/nav_field.o $V/render_fused_$name.o
rm -f $V/render_fused_$name.o
echo built $V/libngp_$name.so
