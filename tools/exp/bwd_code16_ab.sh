# A/B of the 16-bit tile codes on one box: owners + pre-pass per call, byte codes (default) vs nibble codes (PERF_BWD_CODE16=1)
cd /root/repo
for i in 1 2; do
  python tools/exp/bwd_ab.py
  PERF_BWD_CODE16=1 python tools/exp/bwd_ab.py
done
