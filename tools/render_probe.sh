# render kernel time for several builds (MJRL_LIB) and tile targets: tools/render_probe.sh
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for spec in "w0 8" "w0 4" "w5 8" "w5 4"; do
  set -- $spec; lib=$1; tgt=$2
  unset MJRL_LIB; [ $lib != w0 ] && export MJRL_LIB=$GRAFT_REPO_ROOT/tools/ab/libs/lib_$lib.so
  export MJRL_RENDER_TARGET=$tgt
  rm -rf $O/rp_x
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_x -- python3 $GRAFT_REPO_ROOT/tools/render_rate.py 512 > $O/rp_x.log 2>&1
  f=$(find $O/rp_x -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'render_kernel' in r['Name']: print('$lib target $tgt', 'render avg us', float(r['AverageNs'])/1e3)
"
  rm -rf $O/rp_x
done
