#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <name> <extra hipcc flags...> -> exon_amd/lib/libexon_hip_<name>.so
# (select it with EXON_HIP_LIB=exon_amd/lib/libexon_hip_<name>.so; objects go to a scratch directory, the product build is untouched)
set -e
cd "$(dirname "$0")/../exon_amd/csrc"
name=$1; shift
tmp=$(mktemp -d)
for f in kernels.hip gpu_parse.hip bam_parse.hip bcf_parse.hip text_columns.hip capi.cpp stream.cpp scan.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 "$@" -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -c $f -o $tmp/${f%.*}.o &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 "$@" -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -mllvm -structurizecfg-skip-uniform-regions -c inflate.hip -o $tmp/inflate.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 "$@" -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -mllvm -structurizecfg-skip-uniform-regions -c gzip_stream.hip -o $tmp/gzip_stream.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $tmp/*.o -o ../lib/libexon_hip_$name.so -lz -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf $tmp
echo built ../lib/libexon_hip_$name.so
