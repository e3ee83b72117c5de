"""usage: python dev/kres.py REGEX lib.so [lib2.so ...] -- VGPRs / spills / scratch of the matching kernels in each library"""
import os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from kernel_resources import kernel_resources
pat = re.compile(sys.argv[1])
R = {p: kernel_resources(os.path.abspath(p)) for p in sys.argv[2:]}
for k in sorted(R[sys.argv[2]]):
    if pat.search(k):
        print(k[-44:], {os.path.basename(p)[9:-3]: (R[p][k]["vgpr"], R[p][k]["vgpr_spill"], R[p][k]["scratch"]) for p in R if k in R[p]})
