# development build (--dev): the only one that reads DXTEX_* knobs
for o in "0,1,2,3,4,5,6,7,8,9" "0,2,3,4,5,6,7,8,1,9" "5,6,7,8,0,2,3,4,1,9" "0,2,3,4,6,7,8,5,1,9" "0,2,3,4,1,6,7,8,5,9" "9,0,1,2,3,4,5,6,7,8" "0,2,3,4,5,1,6,7,8,9"; do
  echo "order $o"; DXTEX_BC6H_ORDER=$o python tools/r03_probe.py --dev bc6h 2>&1 | grep -E "bc6h 4096|payload|perturb_m[1569] "
done
