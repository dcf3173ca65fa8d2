#!/bin/bash
# A/B of library builds on one convolution shape: tools/abconv.sh "kind B S Cin Cout [res]" lib1 lib2 ...  (default = the shipped library)
shape=$1; shift
for rep in 1 2; do
for lib in "$@"; do
    if [ "$lib" = default ]; then unset LDM_HIP_LIB; else export LDM_HIP_LIB=$PWD/devlibs/lib$lib.so; fi
    timeout -k 10 100 python3 tools/convone.py $shape 2>&1 | grep "TF/s"
done; done
