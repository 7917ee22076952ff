#!/bin/bash
# r05 GPU run 4: the WHOLE -m gpu suite (as the driver runs it, without -x so that every failure shows) + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -120 ) > gpurun_out/r05/run4_pytest.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/r05/run4_smoke.txt 2>&1
tail -15 gpurun_out/r05/run4_pytest.txt; cat gpurun_out/r05/run4_smoke.txt
