#!/bin/bash
# usage (on the GPU box): tools/calib/run_calib.sh   -> gpurun_out/calib/summary.txt  (copy to profiles/r04_counter_calibration.*)
R=$PWD; mkdir -p gpurun_out/calib; rm -rf gpurun_out/calib/*
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/calib/counter_calib.hip -o gpurun_out/calib/counter_calib 2> gpurun_out/calib/build.err || { cat gpurun_out/calib/build.err; exit 1; }
gpurun_out/calib/counter_calib > gpurun_out/calib/plain.txt
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 180 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/calib -o $tag -- $R/gpurun_out/calib/counter_calib > $R/gpurun_out/calib/$tag.out 2> $R/gpurun_out/calib/$tag.err || echo "pass $tag failed (rc $?)" >> $R/gpurun_out/calib/failed.txt
done
cd $R
python - <<'PY'
import csv, glob, collections, re
actual = {}
for l in open('gpurun_out/calib/plain.txt'):
    m = re.match(r"CALIB (\S+)\s+bytes (\d+)\s+ms ([\d.]+)\s+GB/s ([\d.]+)", l)
    if m: actual[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
short = {'rd_stream16': 'rd_stream16', 'rd_gather<5>': 'rd_gather40', 'rd_gatherILi5': 'rd_gather40', 'rd_gather<9>': 'rd_gather72', 'rd_gatherILi9': 'rd_gather72', 'rd_gather<1>': 'rd_gather8', 'rd_gatherILi1': 'rd_gather8', 'wr_stream16': 'wr_stream16',
         'wr_rec<9, 1>': 'wr_rec72', 'wr_recILi9ELi1': 'wr_rec72', 'wr_rec<5, 2>': 'wr_rec40x2', 'wr_recILi5ELi2': 'wr_rec40x2', 'wr_scatter<5>': 'wr_scatter40', 'wr_scatterILi5': 'wr_scatter40'}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/calib/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = next((v for s, v in short.items() if s in r['Kernel_Name']), None)
        if k: vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = open('gpurun_out/calib/summary.txt', 'w')
def P(*a):
    s = ' '.join(str(x) for x in a); print(s); out.write(s + '\n')
P('kernel         actual bytes    GB/s   | counters of the SECOND (warm) launch of each kernel: value, and value x unit / actual bytes')
for k, (b, ms, gbs) in actual.items():
    P('%-14s %13.0f %7.1f' % (k, b, gbs))
    for c, v in sorted(vals[k].items()):
        x = v[-1]
        if c in ('FETCH_SIZE', 'WRITE_SIZE'): P('      %-28s %16.1f KB   -> reported / actual = %.4f' % (c, x, x * 1024 / b))
        else: P('      %-28s %16.0f      -> x32 B / actual = %.4f ; x64 B = %.4f ; x128 B = %.4f' % (c, x, x * 32 / b, x * 64 / b, x * 128 / b))
out.close()
PY
find gpurun_out/calib -name "*.csv" -size +2M -delete; rm -f gpurun_out/calib/counter_calib
