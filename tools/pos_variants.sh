#!/bin/bash
# Developer helper (build container): libmmvae_hip.so variants that differ in the pos_conv ring depths -> moving-mnist-vae_amd/libmmvae_v<i>.so
# usage: tools/pos_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...
set -e
cd "$(dirname "$0")/../moving-mnist-vae_amd/csrc"
make -j8 >/dev/null
i=0
for flags in "$@"; do
  i=$((i+1)); d=/tmp/posvar$i; mkdir -p $d
  for f in conv_pos_a conv_pos_b conv_pos_c; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-variable -fvisibility=hidden $flags -c $f.hip -o $d/$f.o &
  done
  wait
  objs=$(ls ../../build/csrc/*.o | grep -v conv_pos_)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o ../libmmvae_v$i.so $objs $d/conv_pos_a.o $d/conv_pos_b.o $d/conv_pos_c.o -ldl
  echo "variant $i: $flags"
done
