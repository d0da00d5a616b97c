for rep in 1 2 3; do
echo -n "base      "; python tools/bench_conv.py 2>&1 | grep conv | awk '{printf "%s ", $(NF-1)}'; echo
echo -n "256x64 w8 "; DF_CONV_TILE=256064 DF_CONV_W8=7 python tools/bench_conv.py 2>&1 | grep conv | awk '{printf "%s ", $(NF-1)}'; echo
done
