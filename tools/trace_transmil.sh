# usage: tools/trace_transmil.sh tag [ENV=VAL ...]   -> gpurun_out/s2/trace_<tag>.txt (one forward's kernels with start / end in us)
export TMPDIR=/tmp
tag=$1; shift
mkdir -p /root/repo/gpurun_out/s2
rm -rf /tmp/tr_$tag
(cd /tmp && env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o t -- python /root/repo/bench.py --workload transmil --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
python - $f $tag <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'tm_cls_head' in r['Kernel_Name']]
end=idx[-1]; start=idx[-2]+1
t0=int(rows[start]['Start_Timestamp'])
out=open('/root/repo/gpurun_out/s2/trace_%s.txt'%sys.argv[2],'w')
for r in rows[start:end+1]:
    out.write("%9.1f %9.1f %6.1f q=%s  %s\n"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id','?'),r['Kernel_Name'][:48]))
print(sys.argv[2], "forward span us:", (int(rows[end]['End_Timestamp'])-t0)/1e3)
P
