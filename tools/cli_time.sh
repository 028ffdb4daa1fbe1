cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time, subprocess, os
sys.path.insert(0,'tools')
import gen_synth
ia, ib, _ = gen_synth.make_pair(1024, 436, 5)
gen_synth.write_pgm('/tmp/a.pgm', ia); gen_synth.write_pgm('/tmp/b.pgm', ib)
exe='of_dis_amd/lib/run_OF_INT'
for env in ({}, {'OFDIS_CONTRACT':'fused'}, {'AMD_LOG_LEVEL':'0','HIP_ENABLE_DEFERRED_LOADING':'1'}):
    ts=[]
    for r in range(4):
        t0=time.perf_counter(); p=subprocess.run([exe,'/tmp/a.pgm','/tmp/b.pgm','/tmp/o.flo','2'],env=dict(os.environ,**env),capture_output=True,text=True); ts.append(time.perf_counter()-t0)
    print(env, [round(t,3) for t in ts]); 
print(p.stdout[-600:])
# baseline: a process that only initialises HIP and allocates
src=r'''
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sys/time.h>
static double now(){struct timeval tv;gettimeofday(&tv,0);return tv.tv_sec*1e3+tv.tv_usec/1e3;}
__global__ void k(int*p){*p=1;}
int main(){double t0=now(); hipInit(0); double t1=now(); int*d; hipMalloc(&d,4); double t2=now(); hipLaunchKernelGGL(k,1,1,0,0,d); hipDeviceSynchronize(); double t3=now();
printf("hipInit %.1f ms, first hipMalloc %.1f ms, first kernel %.1f ms\n",t1-t0,t2-t1,t3-t2);}
'''
open('/tmp/h.hip','w').write(src)
subprocess.run(['hipcc','--offload-arch=gfx950','-O2','/tmp/h.hip','-o','/tmp/h'],check=True)
for r in range(3):
    t0=time.perf_counter(); p=subprocess.run(['/tmp/h'],capture_output=True,text=True); print(round(time.perf_counter()-t0,3), p.stdout.strip())
PY
