L=/root/repo/semantic_segmentation_amd
for n in 2 4 8; do MFL_RING_NGW=$n MFL_LIBRARY=$L/libmfl_c2exp.so timeout -k 10 100 python3 tools/abl_time.py 2>&1 | grep -v amdgpu.ids || exit 1; done
for lib in c4w3 c4w2; do for n in 1 2; do ABL_SHAPE=19,1024,2048,11,2 MFL_RING_NGW=$n MFL_LIBRARY=$L/libmfl_$lib.so timeout -k 10 100 python3 tools/abl_time.py 2>&1 | grep -v amdgpu.ids || exit 1; done; done
