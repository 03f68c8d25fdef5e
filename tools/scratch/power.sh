cd $GRAFT_REPO_ROOT
ls /sys/class/drm/ | head -30
for c in /sys/class/drm/card*/device; do echo "$c: $(cat $c/gpu_busy_percent 2>/dev/null) busy, hwmon: $(ls $c/hwmon 2>/dev/null | tr '\n' ' ')"; done 2>/dev/null | head -20
( timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-configs > /tmp/b.json 2>/dev/null ) &
BP=$!
sleep 40
for i in 1 2 3 4 5 6; do
for c in /sys/class/drm/card*/device; do
  b=$(cat $c/gpu_busy_percent 2>/dev/null)
  if [ "${b:-0}" -gt 5 ]; then
    h=$(ls -d $c/hwmon/hwmon* | head -1)
    echo "$c busy=$b power_avg=$(cat $h/power1_average 2>/dev/null) power_input=$(cat $h/power1_input 2>/dev/null) cap=$(cat $h/power1_cap 2>/dev/null) sclk=$(cat $h/freq1_input 2>/dev/null) mclk=$(cat $h/freq2_input 2>/dev/null) temp=$(cat $h/temp1_input 2>/dev/null) $(grep '\*' $c/pp_dpm_sclk 2>/dev/null | tr '\n' ' ')"
  fi
done
sleep 0.7
done
wait $BP
head -c 250 /tmp/b.json; echo
