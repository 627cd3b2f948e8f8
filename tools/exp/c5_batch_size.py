#!/usr/bin/env python
"""Config 5, line-local, T = 2^28: the panorama block at other batch sizes (rows_per_batch x 4096 rays as square-ish tiles) and
splits between line-local and tcnn-layout levels."""
import json, sys
sys.path.insert(0, '.')
from perf_amd import panorama as P
for rows, tile, mr in ((4, (128, 128), 64), (16, (256, 256), 64), (16, (256, 256), 200), (32, (256, 512), 200), (8, (128, 256), 200), (4, (128, 128), 200)):
    b = P.render_panorama_block(28, rows_per_batch=rows, layout='line_local', tile=tile, local_min_res=mr)
    r = b['roofline']
    print(json.dumps({'rows_per_batch': rows, 'tile': tile, 'local_min_res': mr, 'seconds_per_panorama': b['seconds_per_panorama'], 'ray_samples_per_s': round(b['ray_samples_per_s'] / 1e9, 3),
                      'encode_ms': r['ms_per_launch'], 'samples_per_launch': r['samples_per_launch'], 'frac': r['frac']}), flush=True)
