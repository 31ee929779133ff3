#!/bin/bash
# scratch: the GPU session of the moment
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-400
