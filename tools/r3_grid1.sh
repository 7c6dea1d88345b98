#!/bin/bash
out=gpurun_out/r3
mkdir -p $out
GRID_MODES=0,2,3,1 timeout -k 10 600 python -u tools/exp_grid.py 8k balls gears refract > $out/grid2.txt 2>&1 || { tail -20 $out/grid2.txt; exit 1; }
grep -v "^\[rt\] reference" $out/grid2.txt
