export LB_SYNTH_CACHE=/tmp
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.2f frames/s  %.2f ms' % (d['value'], d['ms_per_step']))" $1; }
C="--steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary"
for rep in 1 2; do
timeout 300 python bench.py $C --no-materialise > /tmp/t.json 2>/dev/null; echo "lazy frames                      : $(val /tmp/t.json)"
timeout 300 python bench.py $C --materialise-mode after > /tmp/t.json 2>/dev/null; echo "copied + built after the return  : $(val /tmp/t.json)"
timeout 300 python bench.py $C --materialise-mode engine > /tmp/t.json 2>/dev/null; echo "engine (side stream, behind LPIPS): $(val /tmp/t.json)"
done
