#!/bin/bash
# Developer build of a library variant: tools/devbuild.sh NAME "-DFLAG=1 ..." [file.hip ...]  ->  devlibs/libNAME.so
# (the listed sources, default tapconv.hip, are compiled with the flags; the other objects come from build/csrc)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2 || true
files=${@:-tapconv.hip}
mkdir -p build/dev devlibs
objs=""
for f in tapconv.hip attention.hip elementwise.hip batchnorm.hip ldm_api.hip; do
    if [[ " $files " == *" $f "* ]]; then
        o=build/dev/${name}_${f%.hip}.o
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-result \
              -Wno-unused-command-line-argument -DLDM_DEV_BUILD $flags -c diffusionmodel_amd/csrc/$f -o $o 2>&1 | grep -v "not a recognized feature" || true
        test -f $o
        objs="$objs $o"
    else
        objs="$objs build/csrc/${f%.hip}.o"
    fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o devlibs/lib$name.so $objs
echo devlibs/lib$name.so
