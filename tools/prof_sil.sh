#!/bin/bash
OUT=$PWD/gpurun_out/prof_sil
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/tools/exp/sil_time.py 512"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc -o pmc -- $CMD > $OUT/pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN --output-format csv -d $OUT/pmc2 -o pmc -- $CMD > $OUT/pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for name in ('pmc','pmc2'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f'/root/repo/gpurun_out/prof_sil/{name}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:40]
            if 'sil' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
tail -3 $OUT/pmc2.log
