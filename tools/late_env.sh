cd /root/repo
run() { python -c "
import sys
$1
import runpy
sys.argv=['bench.py','--no-cpu-baseline','--no-dreamer-leg','--no-context','--steps','20000','--sustained-seconds','2']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['sustained']['env_steps_per_s'], d['config'].get('kernargs'))"; }
echo "A: env set before torch import"; run "pass"
echo "B: torch imported first"; run "import torch"
echo "C: torch imported, is_available() called first"; run "import torch; torch.cuda.is_available()"
echo "D: torch imported, device_count + a tensor made first"; run "import torch; torch.zeros(1, device='cuda')"
echo "E: explicit 1"; HIP_FORCE_DEV_KERNARG=1 run "pass"
