# A/B of several builds of the library in one session: tools/diag/ab_multi.sh "<bench args>" lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for rep in 1 2; do for L in "$@"; do HIPNLP_LIB_PATH=$PWD/$L python3 bench.py --no-cpu-baseline --no-hessian --no-host $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-55s %.4g knots/s  %.5f ms' % ('$L', d['value'], d['ms_per_step']))"; done; done
