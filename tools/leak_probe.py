"""Developer probe: device memory after rounds of context creation / destruction in five configurations (gray, RGB, stereo +
forward-backward, operating point 3, HD RGB): the free-memory delta must not grow from round to round.
    python tools/leak_probe.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from of_dis_amd import capi
from of_dis_amd.params import oppoint
torch.cuda.set_device(0); capi.check(capi.lib().ofdis_set_device(0))
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
f0 = free()
for rep in range(3):
    for (w,h,noc,opp,mode,fb,n) in ((1024,436,1,2,1,0,64),(1024,436,3,2,1,0,32),(1242,375,1,2,2,1,16),(640,480,1,3,1,1,8),(1920,1080,3,2,1,0,20)):
        for k in range(40):
            p = oppoint(opp,w,h,noc=noc).copy(selectmode=mode, usefbcon=fb)
            b = capi.Batch(p, n); b.close()
    print("round", rep, "free bytes delta", f0 - free(), flush=True)
