echo "== default"; PROBE_TOP=0 python tools/r04_quick.py --dev bc7 2>&1 | grep "\^2\|payload"
echo "== small plan at 4096^2"; DXTEX_BC7_SMALL_BLOCKS=1048576 PROBE_TOP=0 python tools/r04_quick.py --dev bc7 2>&1 | grep "\^2\|payload"
echo "== plan X at 4096^2"; DXTEX_BC7_SMALL_PLAN="16|1,3/14,15,18,7,0,2|24/28/25|26" DXTEX_BC7_SMALL_BLOCKS=1048576 PROBE_TOP=0 python tools/r04_quick.py --dev bc7 2>&1 | grep "\^2\|payload"
echo "== plan Y at 4096^2"; DXTEX_BC7_SMALL_PLAN="16|1/24,14,15,18,7,0|3,2/28/25|26" DXTEX_BC7_SMALL_BLOCKS=1048576 PROBE_TOP=0 python tools/r04_quick.py --dev bc7 2>&1 | grep "\^2\|payload"
