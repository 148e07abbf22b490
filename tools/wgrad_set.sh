for sh in "32 32 256 512 1" "16 128 64 256 2" "16 64 128 512 2" "16 32 256 1024 2" "16 32 256 512 2" "16 64 128 256 2" "32 64 128 256 2"; do python tools/bench_wgrad.py $sh 2>/dev/null; done
