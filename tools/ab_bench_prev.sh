#!/usr/bin/env bash
# same-box A/B of the metric: current engine vs the previous build, interleaved
# needs lightkrylov_amd/liblightkrylov_hip_prev.so: the engine of an earlier commit built in a scratch worktree (git worktree add /tmp/oldsrc <commit>; make -C /tmp/oldsrc/lightkrylov_amd/csrc) and copied there
cd "$GRAFT_REPO_ROOT"
run() { python3 -c "
import sys, runpy
from lightkrylov_amd import _capi
import os
if '$1' == 'prev': _capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), 'liblightkrylov_hip_prev.so')
sys.argv = ['bench.py', '--no-cpu-baseline', '--steps', '2', '--warmup', '1']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); ps = d['roofline']['per_sweep']
        print('$1', round(d['value'], 2), round(d['roofline']['frac'], 4), [round(ps[s]['frac'], 3) for s in ('sweep1', 'sweep2', 'sweep3')])
"; }
for i in 1 2 3; do run cur; run prev; done
