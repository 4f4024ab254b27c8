# alternating runs of builds on batches of 1080p and on the two-launch 8192 x 8192: tools/abc5.sh name...
for r in 1 2; do
  for l in "$@"; do
    lib=""; [ "$l" != product ] && lib=tools/exp/libjpeg_amd_$l.so
    echo -n "$l: "; JPEG_AMD_LIBRARY=$lib python tools/run_c3.py 30 1920 1080 512 2>/dev/null
    echo -n "$l: "; JPEG_AMD_LIBRARY=$lib python tools/run_c3.py 100 1920 1080 64 2>/dev/null
    echo -n "$l (two launches): "; JPEG_AMD_QUAD=0 JPEG_AMD_LIBRARY=$lib python tools/run_c3.py 200 2>/dev/null
  done
done
