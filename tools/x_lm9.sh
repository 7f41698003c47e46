#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
cd $ROOT && timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
run() { local label=$1; shift; timeout -k 10 200 python3 $ROOT/bench.py "$@" --no-cpu-baseline > /tmp/o.json 2>/tmp/e.txt; python3 -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('$label', round(d['value']), d['kernels_alone_us'])"; }
run c5 --config c5 --steps 10 --warmup 3
run c5b --config c5 --steps 10 --warmup 3
run c3 --config c3 --steps 8 --warmup 3
run l44103 --frames 8 --length 44103 --steps 5 --warmup 2
run l44102 --frames 8 --length 44102 --steps 5 --warmup 2
python3 $ROOT/tools/bench_stream.py 128 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('stream', d['frames_per_s'])"
