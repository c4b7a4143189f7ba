#!/bin/bash
# how much of the strong-scaling loss is the per-launch tail?  rank 0 of N on a frame 1x / 2x / 4x / 8x as tall (same width, same
# camera): ms per step divided by the height factor = per-1080p-frame-equivalent cost of a launch that carries several frames
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --verify 0 --steps 200 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$*', '->', j['ms_per_step'], 'ms', j['config']['rays_per_frame'], 'rays', round(j['value'],1), 'Mrays/s')"; }
for n in 8 4 2; do for k in 1 2 4 8; do run --as-rank 0/$n --height $((1080*k)); done; done
