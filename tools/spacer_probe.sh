#!/bin/bash
# Developer tool: per-colour sweep time by the size of the spacer allocated between the colours.
for sp in "$@"; do
  echo -n "spacer $sp GB: "
  ITERS=2 REPS=${REPS:-1} DS_SPACER_GB=$sp timeout -k 10 120 python -u tools/per_launch.py | cut -c1-34 | tr "\n" " "; echo
done
