"""Dev tool (GPU box): where does k_conv3_up_split_pp spend a sample?  Needs tools/_haz/librfuse_ppstamps.so (csrc/build.py build(out=..., extra_flags=['-DRF_PP_STAMPS'])):
s_memtime at the phase borders of every workgroup's 4th sample (waves 0 and 4), medians over the workgroups.  python tools/pp_stamps.py [n=8192]"""
import ctypes, os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
os.environ['RFUSE_LIB'] = str(REPO / 'tools' / '_haz' / 'librfuse_ppstamps.so')
import subprocess
import numpy as np
sys.argv = [sys.argv[0]] + sys.argv[1:]
if '--linear' not in sys.argv:
    sys.argv.append('--pm')
exec(open(REPO / 'tools' / 'pp_bench.py').read().split("if '--save'")[0])
lib = ctypes.CDLL(os.environ['RFUSE_LIB'])
buf = (ctypes.c_ulonglong * (1024 * 2 * 16))()
assert lib.rft_pp_read_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(1024, 2, 16).astype(np.int64)
wgs = min(n, 256)
st = st[:wgs]
names = ['zero acc', 'phase B', 'chunk 0', 'chunk 1', 'chunk 2', 'chunk 3', 'relu + statistics', 'barrier 1', 'channel sums + barrier + triples', 'barrier 3', 'normalise + split + store']
for w in (0, 1):
    d = st[:, w, 1:11] - st[:, w, 0:10]
    tot = st[:, w, 10] - st[:, w, 0]
    print('wave %d: sample %d cycles (median; REFCLK 100 MHz units x clock ratio if s_memtime is not the shader clock)' % (4 * w, np.median(tot)))
    for i in range(10):
        print('   %-28s %8.0f' % (names[i + 1], np.median(d[:, i])))
