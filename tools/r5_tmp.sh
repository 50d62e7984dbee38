for i in 1 2 3; do
SPRING_AMD_LIB=spring_amd/lib/x_old.so timeout 300 python tools/scale_probe.py 100000000,150,65536 100000000,150,65536 2>&1 | grep "^n=" | sed 's/search_ms.*lost/lost/' | tail -1
timeout 300 python tools/scale_probe.py 100000000,150,65536 100000000,150,65536 2>&1 | grep "^n=" | sed 's/search_ms.*lost/lost/' | tail -1
done
