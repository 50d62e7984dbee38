cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { SP_OPTS="$1" timeout 300 python tools/scale_probe.py $2 $2 2>&1 | grep "^n=" | tail -1 | sed "s/unmatched.*dev=/dev=/;s/^/[$1] /"; }
for n in 15000000 25000000 30000000 35000000 70000000; do
run "phases=1" $n,150,0
run "phases=2" $n,150,0
run "phases=2,fused=3" $n,150,0
done
