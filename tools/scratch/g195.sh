cd $GRAFT_REPO_ROOT
run() { timeout 900 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-side-configs 2>/tmp/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], 'G', d['config']['chunk_group'])"; grep -i "error\|out of memory" /tmp/err.log | tail -1 | cut -c1-200; }
for r in 1 2; do
echo "G 98  : $(run)"
echo "G 130 : $(run --chunk-group 130)"
echo "G 195 : $(run --chunk-group 195)"
echo "G 390 : $(run --chunk-group 390)"
done
