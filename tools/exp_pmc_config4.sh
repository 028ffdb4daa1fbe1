#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/pmc_config4; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for lpp in 32 64; do
OFDIS_RGB12_LPP=$lpp timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p$lpp -- python $R/tools/config4_probe.py 16 > $OUT/p$lpp.log 2>&1
f=$(find $OUT/p$lpp -name "*counter_collection.csv" | head -1)
python - $f <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_INSTS_VALU",0))[:6]:
    print(k, "launches", cnt[k], {a:int(b) for a,b in v.items()}, "VALU per wave", int(v.get("SQ_INSTS_VALU",0)/max(v.get("SQ_WAVES",1),1)))
PY
rm -rf $OUT/p$lpp
done
