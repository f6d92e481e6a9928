bash tools/ab_variants.sh run "python tools/r04_quick.py --dev cfg4 bc1 2>&1 | grep 'cfg4 BC3\|bc[123] 4096\|sha'"
