#!/bin/bash
# the same bench lines with the library of an earlier commit (tools/micro/libfmx_old.so, built by hand) and the current one, on ONE box
for rep in 1 2; do for lib in tools/micro/libfmx_old.so ""; do
  echo -n "lib=${lib:-current} : "
  FMX_LIB_PATH=${lib:+$(pwd)/$lib} timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%.1f M/s %.2f us/step; short %.1f M; deepfm %.1f M %.1f us; class surface pinned %.1f M, lists %.2f M' % (d['value']/1e6, d['ms_per_step']*1e3, d['short_run']['median_samples_per_s']/1e6, d['secondary']['deepfm']['value']/1e6, d['secondary']['deepfm']['ms_per_step']*1e3, d['class_surface']['pinned_arrays_samples_per_s']/1e6, d['class_surface']['nested_lists_samples_per_s']/1e6))"
done; done
