cd $GRAFT_REPO_ROOT
timeout 600 python tools/kernel_power.py 1.5 2>&1 | grep -v amdgpu.ids
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-configs 2>/dev/null | python -c "import json,sys; l=sys.stdin.readline(); d=json.loads(l); print(len(l), d['ms_per_step'], d['power'])"
timeout 600 python bench.py --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; l=sys.stdin.readline(); d=json.loads(l); print(len(l), d['value'], d['power'])"
timeout 600 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side-configs 2>/dev/null | python -c "import json,sys; l=sys.stdin.readline(); d=json.loads(l); print(len(l), d['ms_per_step'], d['power'])"
