#!/bin/bash
# HBM bytes of one whole training step, per kernel family.  Run from the repo root on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${ROUND:-5}
O=gpurun_out/r$R
mkdir -p $O
rm -rf $O/pmcs_fetch $O/pmcs_write
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcs_fetch -- python3 profiles/tools/pmc_step.py 3 > $O/pmcs_fetch.log 2>&1; tail -2 $O/pmcs_fetch.log
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcs_write -- python3 profiles/tools/pmc_step.py 3 > $O/pmcs_write.log 2>&1; tail -2 $O/pmcs_write.log
VER=$(python3 -c "import sys; sys.path.insert(0, 'mr-mt3_amd'); from mrmt3 import lib; print(lib.load().mrmt3_version())")
python3 profiles/tools/pmc_step_parse.py $O/pmcs_fetch $O/pmcs_write $O/r0${R}_pmc_step_traffic.txt $O/r0${R}_pmc_step_traffic.json $VER 64
# the raw counter files are large: keep only the table
rm -rf $O/pmcs_fetch $O/pmcs_write
