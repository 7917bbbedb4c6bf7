cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=$GRAFT_REPO_ROOT/blaze_amd/lib/libblaze_hip_nopair.so
b() { timeout 600 python bench.py --no-cpu-baseline --no-ntt --no-extras | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(j['ms_per_step'], j['roofline']['kernel_ms'], j['phases_ms']['sort_ms'], j['result_check']['ok'])"; }
for i in 1 2 3; do echo "== bench paired"; b; echo "== bench unpaired"; BLAZE_HIP_LIB=$N b; done
