#!/bin/bash
for lib in "$@"; do echo "=== $lib"; CASAPOSE_HIP_LIB=$PWD/variants/lib_$lib.so python tools/layer_times.py 2>&1 | tail -36; done
