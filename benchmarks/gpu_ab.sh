cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for T in 0 8 12 16 24 32; do echo "== budget $T"; FZ_CHUNKS_PER_WAVE=$T timeout 300 python benchmarks/ab_scan.py 1024 200; done
 echo "== budget 16 flags 2";  FZ_SCAN_FLAGS=2 FZ_CHUNKS_PER_WAVE=16 timeout 300 python benchmarks/ab_scan.py 1024 200
 echo "== budget 16 flags 4";  FZ_SCAN_FLAGS=4 FZ_CHUNKS_PER_WAVE=16 timeout 300 python benchmarks/ab_scan.py 1024 200
 FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/r1/libfzhip_r1.so timeout 300 python benchmarks/ab_scan.py 1024 200) 2>&1 | tee gpurun_out/ab4.log | cut -c1-250
