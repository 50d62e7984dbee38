for lib in "" spring_amd/lib/x_lw7.so "" spring_amd/lib/x_lw7.so; do
  echo "== lib=${lib:-default}"
  SPRING_AMD_LIB=$lib timeout 200 python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,0 2>&1 | tail -1
  SPRING_AMD_LIB=$lib timeout 200 python tools/scale_probe.py 20000000,150,0,10000,gen,25 20000000,150,0,10000,gen,25 2>&1 | grep "^n=" | tail -1 | sed 's/search_ms.*lost/lost/'
done
