#!/bin/bash
# kernel trace of the driver at the reference's default batch: are the slow outer batches slow KERNELS or GAPS between them?
OUT=$PWD/gpurun_out/trace_b256
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python3 $ROOT/main.py --batch_size 256 --synthetic_batches 12 --synthetic --smpl_dir /nonexistent --j_regressor_init /nonexistent > $OUT/run.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob
f = glob.glob('**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]) for r in csv.DictReader(open(f))]
rows.sort()
# segments: runs of k_chain_bwd-terminated iterations; split the trace at gaps > 2 ms (outer-batch boundaries)
segs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 2_000_000: segs.append(cur); cur = []
    cur.append(b)
segs.append(cur)
for s in segs:
    if len(s) < 500: continue
    span = (s[-1][1] - s[0][0]) / 1e6; busy = sum(e - st for st, e, _ in s) / 1e6
    gaps = sorted(((b[0] - a[1]) / 1e3 for a, b in zip(s, s[1:])), reverse=True)
    print('segment: %5d kernels  span %7.2f ms  kernel time %7.2f ms  largest gaps (us): %s  median gap %.1f us' % (len(s), span, busy, [round(g) for g in gaps[:4]], gaps[len(gaps) // 2]))
PY
