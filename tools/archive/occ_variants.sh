cp srba_amd/lib/libsrba_hip.so /tmp/orig.so
for v in base occ2; do echo "== $v"; if [ $v != base ]; then cp srba_amd/lib/variants/libsrba_hip_$v.so srba_amd/lib/libsrba_hip.so; touch srba_amd/lib/libsrba_hip.so srba_amd/lib/libsrba_engine.so; fi
for k in stereo mono; do timeout 600 python tools/diag_family.py $k 2>&1 | tail -1 | cut -c1-200; done
timeout 600 python bench.py --workload cfg3 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cfg3', d['value'], d['ms_per_step'])"
done; cp /tmp/orig.so srba_amd/lib/libsrba_hip.so
