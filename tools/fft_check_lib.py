import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import parity_checks as P
lib=sys.argv[1]
for N in (1024,2048):
    P.check_fft_plugin(lib, N, count=37)
print("fft plugin parity ok", lib)
