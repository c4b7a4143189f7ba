#!/bin/bash
# beam pre-pass A/B in the throughput configuration: the coarse pass is computed once per dispatch (a batch shares its camera)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_beam_ab; mkdir -p $O
for cam in K1 K2 K0; do for B in 1 4 8; do for beam in 0 1; do
  python bench.py --camera $cam --batch $B --beam $beam --steps 1600 --cpu-seconds 0 2>/dev/null | tail -1 > $O/${cam}_B${B}_beam${beam}.json
  python -c "
import json; j=json.load(open('$O/${cam}_B${B}_beam${beam}.json')); print('$cam batch $B beam $beam:', j['value'], 'Mrays/s', j['ms_per_step'], 'ms', j['verified'], j['config']['iterations_per_ray'], 'it/ray', j['config']['alg_bytes_per_ray'], 'B/ray')"
done; done; done
