#!/usr/bin/env python
"""Which kernels of libperf_hip.so carry scratch (private segment) and how many vector registers they take: compiles every unit of
perf_amd/csrc with the build's flags + -save-temps into a temporary directory and reads the kernel descriptors of the gfx950 assembly
(no GPU needed).  `python tools/scratch_report.py [out.json]`."""
import glob, json, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from perf_amd import build as B


def main():
    tmp = tempfile.mkdtemp(prefix='perf_scratch_')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    procs = []
    for src in sorted(glob.glob(os.path.join(B.CSRC, '*.hip'))):
        unit = os.path.splitext(os.path.basename(src))[0]
        procs.append(subprocess.Popen([hipcc] + B.FLAGS + ['-save-temps', '-c', src, '-o', os.path.join(tmp, unit + '.o')], cwd=tmp,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    for p in procs:
        p.wait()
    filt = shutil.which('c++filt')
    rep = {'flags': B.FLAGS, 'kernels_with_scratch': {}, 'kernels': 0}
    for f in sorted(glob.glob(os.path.join(tmp, '*-hip-amdgcn-amd-amdhsa-gfx950.s'))):
        unit = os.path.basename(f).split('-hip-')[0]
        for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', open(f).read()):
            rep['kernels'] += 1
            if int(m.group(2)) > 0:
                name = subprocess.run([filt, m.group(1)], capture_output=True, text=True).stdout.strip() if filt else m.group(1)
                rep['kernels_with_scratch'].setdefault(unit, []).append({'kernel': re.sub(r'\(.*', '', name), 'scratch_bytes_per_lane': int(m.group(2)),
                                                                        'vgprs': int(m.group(3))})
    shutil.rmtree(tmp, ignore_errors=True)
    txt = json.dumps(rep, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(txt + '\n')


if __name__ == '__main__':
    main()
