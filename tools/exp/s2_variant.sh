#!/bin/bash
# builds tools/ab/libpar_s2<tag>.so = the library with sinc2.hip compiled under extra flags:  s2_variant.sh <tag> <flags...>
set -e
cd "$(dirname "$0")/../.."
tag=$1; shift
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c pyaudiorestoration_amd/csrc/sinc2.hip -o /tmp/sinc2_$tag.o 2>/dev/null
objs=$(ls pyaudiorestoration_amd/csrc/obj/*.o | grep -v '/sinc2.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/sinc2_$tag.o -o tools/ab/libpar_s2$tag.so
echo tools/ab/libpar_s2$tag.so
