#!/bin/bash
# A/B of environment switches on the reference-minibatch step (B = 26, bf16): bash profiles/scripts/b26_ab.sh "VAR=val ..." ...
run() { echo -n "$1: "; env $1 python3 profiles/scripts/graph_b26.py 2>/dev/null | grep "B =  26" | sed 's/, graph.*//'; }
run "X=1"
for e in "$@"; do run "$e"; done
