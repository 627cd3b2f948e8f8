"""Fold the CSVs of tools/exp/mlp_pmc.sh: per MLP kernel, mean counter value per launch and the kernel-trace average duration."""
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]


def short(name):
    m = re.search(r'(mlp_\w+)<perf::(\w+), (\d), (\d)(?:, (\w+))?>', name)
    return f'{m.group(1)}<{m.group(2)},{m.group(3)},{m.group(4)}{"," + m.group(5) if m.group(5) else ""}>' if m else None


out = {}
for f in glob.glob(os.path.join(root, '*', '**', '*counter_collection.csv'), recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        k = short(row['Kernel_Name'])
        if k is None:
            continue
        a = acc.setdefault((k, row['Counter_Name']), [0.0, set()])
        a[0] += float(row['Counter_Value']); a[1].add(row['Dispatch_Id'])
    for (k, c), (v, ids) in acc.items():
        out.setdefault(k, {})[c] = round(v / max(1, len(ids)), 1)
for f in glob.glob(os.path.join(root, 'kt', '**', '*kernel_stats.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row['Name'])
        if k:
            out.setdefault(k, {})['avg_us'] = round(float(row['AverageNs']) / 1e3, 2)
            out[k]['calls'] = int(row['Calls'])
print(json.dumps(out, indent=1, sort_keys=True))
