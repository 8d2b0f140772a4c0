# cfg 4 / cfg 5: FETCH_SIZE / WRITE_SIZE per kernel, one context (looking for write amplification like k_triy_chns's)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for CFG in 4 5; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$CFG_$C
    ACF_HIP_SCALES_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_${CFG}_$C -o pmc --output-format csv -- python bench.py --config $CFG --contexts 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-latency --no-repeats > /tmp/pmc_${CFG}_$C.log 2>&1
  done
  python - $CFG <<'PY'
import csv,sys,collections,glob
cfg=sys.argv[1]
def load(c):
    f=glob.glob('/tmp/pmc_%s_%s/**/*counter_collection.csv'%(cfg,c),recursive=True)[0]
    d=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']==c:
            k=r['Kernel_Name'].split('(')[0][:70]; d[k]+=float(r['Counter_Value']); n[k]+=1
    return d,n
fe,nf=load('FETCH_SIZE'); wr,nw=load('WRITE_SIZE')
print('cfg',cfg,'KB summed over the run (3 launches per kernel form): kernel, launches, FETCH_SIZE MB (uncorrected), WRITE_SIZE MB')
for k in sorted(set(fe)|set(wr), key=lambda k:-(fe[k]+wr[k]))[:14]:
    print('%-72s %4d %9.1f %9.1f'%(k,nf[k],fe[k]/1e3,wr[k]/1e3))
PY
done
