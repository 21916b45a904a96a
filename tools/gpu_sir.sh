#!/bin/bash
# Monte-Carlo parity tests, then the sir_torch benchmark (bit-exactness against the oracle is part of its output)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k sir 2>&1 | tail -3 || exit 1
timeout -k 10 300 python tools/bench_sir.py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['case'], round(d['gpu_s']*1e3,2), 'ms', '%.3g' % d['gpu_edge_visits_per_s'], d['bit_exact_vs_oracle'])"
