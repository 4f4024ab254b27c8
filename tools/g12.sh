cd $GRAFT_REPO_ROOT
for cap in 768 683 700 640 586 512; do echo -n "cap $cap: "; JA_GRID_CAP=$cap JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_prof.so python tools/run_c3.py 300 2>/dev/null; done
for cap in 768 683; do echo -n "cap $cap: "; JA_GRID_CAP=$cap JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_prof.so python tools/run_c3.py 300 2>/dev/null; done
