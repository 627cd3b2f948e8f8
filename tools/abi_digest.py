#!/usr/bin/env python
"""Digest of the C ABI declared in include/perf_hip.h: sha256 over the comment-free, whitespace-normalised text of every
`perf_*` prototype, struct and #define, in file order.  `python tools/abi_digest.py` prints {version, digest};
`--write` records it in include/perf_hip.abi.json.  tests/test_cpu_oracle.py fails when the digest of the header differs
from the recorded one while PERF_ABI_VERSION is unchanged: every signature change must bump the version (the load-time
check of perf_amd/_lib.py can then catch a stale libperf_hip.so)."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'perf_hip.h')
RECORD = os.path.join(ROOT, 'include', 'perf_hip.abi.json')


def digest(path=HEADER):
    text = open(path).read()
    version = int(re.search(r'#define\s+PERF_ABI_VERSION\s+(\d+)', text).group(1))
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    text = re.sub(r'#define\s+PERF_ABI_VERSION\s+\d+', ' ', text)
    text = re.sub(r'\s+', ' ', text).strip()
    return {'version': version, 'digest': hashlib.sha256(text.encode()).hexdigest()}


if __name__ == '__main__':
    d = digest()
    if '--write' in sys.argv:
        json.dump(d, open(RECORD, 'w'), indent=1)
        open(RECORD, 'a').write('\n')
    print(json.dumps(d))
