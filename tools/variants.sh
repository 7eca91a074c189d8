#!/bin/bash
# time several experimental builds of the HIP library in one GPU session (development aid): tools/variants.sh <reads> <lib>...
N=$1; shift
for rep in 1 2; do
  for L in "$@"; do
    echo "-- $(basename $L)"; OATK_HIP_LIB=$L python tools/kbench.py --reads $N --steps 5 2>&1 | tail -2 | head -1
  done
done
