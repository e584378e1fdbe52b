cd benchmarks
echo "== attn fwd trace"; for ns in 1 4; do KT_NW=4 KT_NS=$ns ./ktrace 1024 1280 1280 lfwd | tail -1; done
echo "== attn bwd trace"; for ns in 1 4; do KT_NW=4 KT_NS=$ns ./ktrace 1024 1280 1280 lbwd | tail -1; done
echo "== timing"
for sh in "1024 1280 1280" "1024 1280 10240" "1024 5120 1280" "4096 640 640" "77 2048 1280"; do
 for m in lfwd lbwd; do for ns in 1 2 4 8; do for nw in 4 8; do echo -n "$sh $m ns=$ns nw=$nw: "; KT_TIME=1 KT_NS=$ns KT_NW=$nw ./ktrace $sh $m | tail -1 | cut -c1-9; done; done; done
 for cv in 1 2 4; do for sp in 2 4 8 16 32; do echo -n "$sh ltn cv=$cv split=$sp: "; KT_TIME=1 KT_CV=$cv KT_SPLIT=$sp ./ktrace $sh ltn | tail -1 | cut -c1-9; done; done
done
