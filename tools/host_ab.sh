#!/bin/bash
# host-bound case (64x64, batch 2) under different settings, one box
cd "$(dirname "$0")/.."
for s in "$@"; do for rep in 1 2; do echo -n "$s: "; env $s python tools/host_profile.py --dtype f32 --steps 20 /dev/null 2>&1 | grep "host issue"; done; done
