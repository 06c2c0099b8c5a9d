#!/bin/bash
# fmx_mlp_section (3 x 256, B = 4096) against the workgroup budget of the weight-gradient launch and the place of the split-K reduction
for wgs in 512 256 128 64; do for red in 1 0; do echo -n "FMX_WGRAD_WGS=$wgs FMX_WGRAD_REDUCE=$red : "; FMX_WGRAD_WGS=$wgs FMX_WGRAD_REDUCE=$red timeout -k 10 200 python tools/mlp_section_times.py 2>/dev/null | grep "mlp_chain=1" | cut -c1-60; done; done
