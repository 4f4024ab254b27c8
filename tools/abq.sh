export JPEG_AMD_QUAD=1
for r in 1 2 3; do
  for l in "" qb2end; do echo -n "lib=${l:-product} "; JPEG_AMD_LIBRARY=${l:+tools/exp/libjpeg_amd_$l.so} python tools/run_c3.py 300 2>/dev/null; done
  echo -n "two-launch  "; JPEG_AMD_QUAD=0 python tools/run_c3.py 300 2>/dev/null
done
for l in "" qb2end; do echo -n "lib=${l:-product} "; JPEG_AMD_LIBRARY=${l:+tools/exp/libjpeg_amd_$l.so} python tools/run_c3.py 300 4096 4096 1 2>/dev/null; done
echo -n "strip420    "; JPEG_AMD_QUAD=0 python tools/run_c3.py 300 4096 4096 1 2>/dev/null
