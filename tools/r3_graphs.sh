#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/graphs; mkdir -p $O; cd $R
for gb in 768 1536; do for g in off on off on; do timeout -k 10 300 python3 bench.py --global-batch $gb --graphs $g --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/b.json 2> $O/b.err || tail -3 $O/b.err; python3 -c "import json;d=json.load(open('$O/b.json'));print($gb, '$g', d['ms_per_step'],d['value'])"; done; done
