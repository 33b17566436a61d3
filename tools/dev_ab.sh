# Developer aid: A/B of the whole step between two builds of the library on ONE box (boxes differ by ~2 %), alternating runs:
#   bash tools/dev_ab.sh build/dbg/libbase.so facenet_amd/lib/libfacenet_hip.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do for L in $A $B; do
  python tools/dev_withlib.py $L bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', d['ms_per_step'], d['ms_per_step_train_only'], d['sum_kernel_ms_insitu'])"
done; done
