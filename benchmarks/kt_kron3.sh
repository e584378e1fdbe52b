#!/bin/bash
# LoKr kron3 kernel: eager launch time and one-workgroup phase timeline per column-tile width (NI) and x path (GM)
cd "$(dirname "$0")"
for sh in "1024 1280 1280" "1024 1280 10240" "1024 5120 1280" "4096 640 640" "77 2048 1280"; do
  for m in fwd bwd; do
    for ni in 1 2 4; do for gm in 3 0; do
      echo -n "$sh $m NI=$ni GM=$gm: "; KT_NI=$ni KT_GM=$gm KT_TIME=1 ./ktrace $sh $m | tail -1 | cut -c1-9
    done; done
  done
done
echo "== trace 1024 1280 1280 fwd NI=1 GM=3"; KT_NI=1 ./ktrace 1024 1280 1280 fwd | tail -2
echo "== trace 1024 1280 1280 fwd NI=1 GM=0"; KT_NI=1 KT_GM=0 ./ktrace 1024 1280 1280 fwd | tail -2
