#!/bin/bash
for t in 1 2 3 4 5 6; do echo "=== tile $t"; python tools/layer_times.py --tile $t 2>&1 | tail -36; done
