#!/bin/bash
# A/B of two builds of the HIP library in one GPU session (development aid): tools/ab.sh <old.so> [reads]
# alternates the two libraries three times so that box-to-box and run-to-run drift shows.
OLD=$1; N=${2:-200000}
for i in 1 2 3; do
  echo "-- old"; OATK_HIP_LIB=$OLD python tools/kbench.py --reads $N --steps 5 2>&1 | tail -2
  echo "-- new"; python tools/kbench.py --reads $N --steps 5 2>&1 | tail -2
done
