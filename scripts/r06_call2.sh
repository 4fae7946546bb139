#!/bin/bash
# Round 6, second GPU call: the repro of the hunt's two device-side events, the full suite, then the
# child-histogram experiments and counters.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06_repro
mkdir -p $O
timeout 600 python tests/tools/repro_first_tree.py 0 2 --procs 8 --iters 120 > $O/cfg2_p8.txt 2>&1
timeout 600 python tests/tools/repro_first_tree.py 0 248 --procs 8 --iters 120 > $O/cfg248_p8.txt 2>&1
timeout 400 python tests/tools/repro_first_tree.py 0 2 --procs 1 --iters 400 > $O/cfg2_p1.txt 2>&1
timeout 400 python tests/tools/repro_first_tree.py 0 2 --procs 8 --iters 120 --no-drain > $O/cfg2_p8_nodrain.txt 2>&1
tail -5 $O/*.txt
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_pytest_a.txt 2>&1; tail -15 gpurun_out/r06_pytest_a.txt )
bash scripts/r06_child.sh all > gpurun_out/r06_child_all.txt 2>&1
tail -60 gpurun_out/r06_child_all.txt
