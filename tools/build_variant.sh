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
# every object of the product library except render_fused.o, which the variant replaces (field_train.hip shares ngp_field.h: pass FT=1 to rebuild it with the same flags)
OBJS=$(ls ../lib/obj/*.o | grep -v render_fused.o)
if [ -n "$FT" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-strict-aliasing -Wno-unused-function \
      -DNGP_BUILD "$@" -c field_train.hip -o $V/field_train_$name.o
  OBJS="$(echo $OBJS | tr ' ' '\n' | grep -v field_train.o) $V/field_train_$name.o"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libngp_$name.so $OBJS $V/render_fused_$name.o
rm -f $V/render_fused_$name.o
echo built $V/libngp_$name.so
