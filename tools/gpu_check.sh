#!/bin/bash
# One GPU-box call: the -m gpu tests, three driver-style 20-step bench runs, one default bench run.
#   bash tools/gpu_check.sh <tag> [pytest -k expression]
set -o pipefail
tag=${1:-run}
sel=${2:-}
out=gpurun_out
mkdir -p $out
if [ -n "$sel" ]; then
  python -m pytest tests -m gpu -x -q -k "$sel" 2>&1 | tee $out/${tag}_pytest.log | tail -6 || exit 1
else
  python -m pytest tests -m gpu -x -q 2>&1 | tee $out/${tag}_pytest.log | tail -6 || exit 1
fi
rm -f $out/${tag}_bench20.json
for i in 1 2 3; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tee -a $out/${tag}_bench20.json | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20-step run: %.1f M samples/s, %.2f us/step' % (d['value']/1e6, d['ms_per_step']*1e3))" || exit 1
done
timeout -k 10 400 python bench.py 2>$out/${tag}_bench.err | tee $out/${tag}_bench.json | \
  python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('default run: %.1f M samples/s, %.2f us/step' % (d['value']/1e6, d['ms_per_step']*1e3))
print('short_run', d.get('short_run'))
k=d['roofline'].get('kernels',{})
print('kernels us: fwd %.2f upd %.2f sort %.2f' % (k['k_fm_forward']['avg_launch_ms']*1e3, k['k_fm_update']['avg_launch_ms']*1e3, k['k_sort_occ']['avg_launch_ms']*1e3))
print('roofline frac %.3f of 8 TB/s, %.3f of measured stream read %.0f GB/s' % (d['roofline']['frac'], d['roofline']['frac_of_measured_stream_read'], d['measured_stream_read_GBps']))
print('chunks', d.get('step_us_over_100_step_chunks'))
s=d.get('secondary',{}).get('deepfm',{})
print('deepfm', s.get('value'), s.get('ms_per_step'), s.get('error'))
print('class_surface', d.get('class_surface'))
"
