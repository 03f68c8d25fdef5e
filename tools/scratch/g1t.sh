cd $GRAFT_REPO_ROOT
for e in 0 2 1 3; do
echo "=== EXP=$e"
FB_C1G=1 FB_C1G_EXP=$e timeout 120 tools/scratch/g1_trace.bin 1024 256 401408 2>&1 | awk '/kernel/ {print} /group/ {g=$2} /step/ { if (NR>0) { n[g]++; for (k=3;k<=9;k++) v[g,k]=$k; if (prev[g]!="") { split(prev[g],a," "); d[g,1]+=v[g,3]-a[9]; } d[g,2]+=$4-$3; d[g,3]+=$5-$4; d[g,4]+=$6-$5; d[g,5]+=$7-$6; d[g,6]+=$8-$7; d[g,7]+=$9-$8; prev[g]=$0; if (first[g]=="") first[g]=$3; last[g]=$9 } } END { for (g in n) { printf "group %s: steps %d, ticks/step %.0f : top->waited %.0f | ->barrier %.0f | ->dma %.0f | ->reads+half0 %.0f | ->mid barrier %.0f | ->half1 %.0f | ->next top %.0f\n", g, n[g], (last[g]-first[g])/n[g], d[g,2]/n[g], d[g,3]/n[g], d[g,4]/n[g], d[g,5]/n[g], d[g,6]/n[g], d[g,7]/n[g], d[g,1]/(n[g]-1) } }'
done
