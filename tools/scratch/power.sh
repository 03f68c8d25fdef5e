cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -i "power\|sclk\|mclk" | head -8
( timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-configs > /tmp/b.json 2>/dev/null ) &
BP=$!
sleep 45
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "Average Graphics Package Power\|Current Socket\|sclk\|power (W)" | tr '\n' ' ' | cut -c1-300; echo
  sleep 0.5
done
wait $BP
head -c 300 /tmp/b.json; echo
( timeout 300 python bench.py --model resnet152 --stem standard --pixels 224 --images 2048 --steps 6 --warmup 1 --no-cpu-baseline --no-kernel-timing > /tmp/b2.json 2>/dev/null ) &
BP=$!
sleep 50
for i in $(seq 1 8); do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "Average Graphics Package Power\|Current Socket\|sclk\|power (W)" | tr '\n' ' ' | cut -c1-300; echo
  sleep 0.5
done
wait $BP
head -c 200 /tmp/b2.json; echo
