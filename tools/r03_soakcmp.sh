#!/bin/bash
# is the host-RSS growth of a 12 000-frame soak new?  the same script on the end-of-round-2 sources (archive shipped in gpurun_out-free path) and on this tree
tag=${1:-r03sk}; mkdir -p gpurun_out/$tag
mkdir -p /tmp/r02 && tar -xzf tools/bin/r02_src.tgz -C /tmp/r02
(cd /tmp/r02 && make -C lumenrenderer_amd/csrc -j8 > /dev/null 2>&1; make -C oracle -s; timeout 900 python tools/soak.py 12000 2>&1 | tail -4) | sed 's/^/r02: /' | tee gpurun_out/$tag/soak_r02.txt
timeout 900 python tools/soak.py 12000 2>&1 | tail -4 | sed 's/^/r03: /' | tee gpurun_out/$tag/soak_r03.txt
timeout 900 python tools/soak.py 24000 2>&1 | tail -4 | sed 's/^/r03 24000: /' | tee -a gpurun_out/$tag/soak_r03.txt
