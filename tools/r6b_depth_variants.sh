mkdir -p gpurun_out/r6b; cd $GRAFT_REPO_ROOT
export DEPTH_NOBATCH=1 DEPTHS=1,2,3,4,5,6,8
run() { tag=$1; shift; timeout 300 python tools/depth_probe.py 64 fused "$@" > gpurun_out/r6b/$tag.json 2> gpurun_out/r6b/$tag.err; echo "$tag rc=$?"; }
run xcu
GPU_MAX_HW_QUEUES=8 run xcu_q8
GPU_MAX_HW_QUEUES=2 run xcu_q2
run mode2 fused_xcu_max=0
run mode1 fused_xcu_max=0 fused_split=0
run mode0 fused_xcu_max=0 fused_mw_max=0
GPU_MAX_HW_QUEUES=8 run mode1_q8 fused_xcu_max=0 fused_split=0
GPU_MAX_HW_QUEUES=8 run mode0_q8 fused_xcu_max=0 fused_mw_max=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6b/*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'ERR', e); continue
    print(f, {k:(v['ms_per_pass'], int(v['frames_per_s']//1000), v['bit_identical_to_D1']) for k,v in d['fused'].items()})
PY
