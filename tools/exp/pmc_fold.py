"""Dev tool: average every counter of rocprofv3 --pmc csv outputs per kernel name.  python tools/exp/pmc_fold.py <dir> [substr]"""
import csv, glob, json, sys
from collections import defaultdict
tot, cnt = defaultdict(float), defaultdict(int)
sub = sys.argv[2] if len(sys.argv) > 2 else 'hashgrid_fwd'
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if sub in r['Kernel_Name']:
            k = (r['Kernel_Name'].split('(')[0].split('<')[0][-40:], r['Counter_Name'])
            tot[k] += float(r['Counter_Value']); cnt[k] += 1
print(json.dumps({f'{k[0]}:{k[1]}': round(tot[k] / cnt[k], 1) for k in sorted(tot)}, indent=0))
