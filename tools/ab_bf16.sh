# A/B variants of libsr_hip.so that differ in conv_bf16.hip compile-time switches:  bash tools/ab_bf16.sh <tag> "<-D flags>"
set -e
cd "$(dirname "$0")/../image_restoration_amd/csrc"
make -j8 > /dev/null
T=$(mktemp -d)
for f in *.hip; do
  if [ "$f" = conv_bf16.hip ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $2 -c $f -o $T/${f%.hip}.o
  else
    cp build/${f%.hip}.o $T/
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libsr_hip_$1.so $T/*.o
rm -rf $T
