#!/bin/bash
# usage: tools/variant_layers.sh "<tiles>" variant...
tiles=$1; shift
for lib in "$@"; do for t in $tiles; do echo "=== $lib t$t"; CASAPOSE_HIP_LIB=$PWD/variants/lib_$lib.so python tools/layer_times.py --tile $t 2>&1 | tail -36; done; done
