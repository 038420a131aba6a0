#!/bin/bash
# usage: tools/sweep_tiles.sh "<tiles>" [extra layer_times args]
tiles=$1; shift
for t in $tiles; do echo "=== tile $t"; python tools/layer_times.py --tile $t "$@" 2>&1 | tail -36; done
