#!/bin/bash
# What bounds tv_fused<3,true,0>?  Timing-only library variants (tools/ab_build.py) on the 4096-pair workload:
#   alias        all wavefronts work on the first 8 frame groups' memory (operands L2 resident)
#   noload       no VMEM loads in the step loop
#   nodiv        quotient = one multiplication, root = v_sqrt alone
#   noload_nodiv both
# plus the VALU issue probe.   gpurun -- "bash tools/exp_fused_bound.sh"   ->  gpurun_out/exp_fused_bound/
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/exp_fused_bound; mkdir -p $OUT
if [ -x tools/probes/issue_probe ]; then timeout 120 tools/probes/issue_probe > $OUT/issue_probe.txt 2>&1; fi
for v in base alias noload nodiv noload_nodiv; do
  lib=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so; [ $v = base ] && lib=$R/of_dis_amd/lib/libofdis_hip.so
  [ -f $lib ] || continue
  for p in 1; do
    echo -n "$v pipeline=$p : "; timeout 300 python tools/kbench.py OFDIS_LIB=$lib -- --steps 10 --warmup 3 --no-extras --pipeline $p
  done
done 2>&1 | tee $OUT/variants.txt
cat $OUT/issue_probe.txt
