cd benchmarks
echo "== library plan"; timeout 200 ./lcbench | grep -v "^#" | cut -c1-40,78-200
for ns in 2 3 5 8; do echo "== NS=$ns"; LCB_NS=$ns timeout 100 ./lcbench attn | grep -v "^#" | cut -c1-40,78-200; done
echo "== sd15 MI1 NW8"; LCB_MI=1 LCB_NW=8 timeout 100 ./lcbench sd15_320 | grep -v "^#" | cut -c1-40,78-200
echo "== sd15 MI2 NW8"; LCB_MI=2 LCB_NW=8 timeout 100 ./lcbench sd15_320 | grep -v "^#" | cut -c1-40,78-200
echo "== sd15 MI1 NW4"; LCB_MI=1 LCB_NW=4 timeout 100 ./lcbench sd15_320 | grep -v "^#" | cut -c1-40,78-200
echo "== ff NS4";  LCB_NS=4 LCB_D2MAX=10 timeout 100 ./lcbench ffup1280 | grep -v "^#" | cut -c1-40,78-200
echo "== ff NS20";  LCB_NS=20 timeout 100 ./lcbench ffup1280 | grep -v "^#" | cut -c1-40,78-200
