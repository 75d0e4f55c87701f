#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tests/hw/final_profile.sh r2_v2 2>&1 | tail -40
