cd $GRAFT_REPO_ROOT
for v in old new old new; do
  export GPK_LIB_PATH=$GRAFT_REPO_ROOT/geopolars_amd/variants/$v.so
  timeout 300 python tools/bench_ops.py 2>&1 | grep -a '"op"' | python -c "
import sys,json
out=[]
for l in sys.stdin:
    d=json.loads(l)
    if d['op'] in ('area','bounds','centroid','euclidean_length') and ('64-vertex' in d['workload'] or 'power' in d['workload']): out.append('%s %.2f' % (d['op'][:4], d['GBps']/1000))
print('$v', ' | '.join(out))
"
done
