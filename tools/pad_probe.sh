#!/bin/bash
# Developer tool: per-colour sweep time for the first allocation of a fresh process,
# by plane padding (outer list, env SPADS) and colour padding (arguments), both in MB.
for sp in $SPADS; do for cp in "$@"; do
  echo -n "spad $sp cpad $cp: "
  ITERS=2 REPS=1 DS_PLANE_PAD=$((sp*1048576)) DS_COLOUR_PAD=$((cp*1048576)) timeout -k 10 120 python -u tools/per_launch.py | cut -c9-30
done; done
