#!/bin/bash
# waves per workgroup of k_fm_forward / k_fm_update (FMX_WPB_FWD / FMX_WPB_UPD) against the online loop
for f in 1 2 4; do for u in 1 2 4; do echo -n "fwd $f upd $u : "; FMX_WPB_FWD=$f FMX_WPB_UPD=$u timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --loop-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f M/s %.2f us/step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done; done
