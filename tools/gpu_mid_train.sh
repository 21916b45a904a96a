#!/bin/bash
# training-step kernel averages on mid-size shapes (HIP-event profiler of the library, tools/train_75k.py)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $R
for S in "1893 13835 1 20" "1893 13835 8 10" "7066 100736 1 10" "7066 100736 4 6" "75000 500000 4 3"; do
  python tools/train_75k.py $S 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$S'.ljust(22), 'step_ms %.3f' % d['ms_per_step'], 'fwd_us %.1f' % d['fwd_step_kernel_avg_us'], 'bwd_us %.1f' % d['bwd_interval_kernel_avg_us'])"
done
