cd /tmp && export TMPDIR=/tmp
for s in 1 2 3 4 0; do
  SNK_ORB_FAST_STOP=$s rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fastphase/s$s -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --ba-windows 0 --pose-frames 0 --track-frames 0 --gba-keyframes 0 --frame-calls 0 --kitti-steps 0 > /dev/null 2>&1
done
python - <<PY
import csv,glob,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/fastphase'
for s in (1,2,3,4,0):
    acc={}
    for f in glob.glob(f'{root}/s{s}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'fast_kernel' in r['Kernel_Name']:
                acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    m={k:sum(v)/len(v) for k,v in acc.items()}
    print('stop',s,{k:round(m[k]/m['SQ_WAVES'],1) for k in m if k!='SQ_WAVES'})
PY
