# alternating runs of builds of the library on QUAD-shaped images: tools/abq.sh name... (tools/exp/libjpeg_amd_<name>.so; "product" = the product build)
for r in 1 2 3; do
  for l in "$@"; do
    lib=""; [ "$l" != product ] && lib=tools/exp/libjpeg_amd_$l.so
    echo -n "$l: "; JPEG_AMD_LIBRARY=$lib python tools/run_c3.py 300 2>/dev/null
  done
done
for l in "$@"; do
  lib=""; [ "$l" != product ] && lib=tools/exp/libjpeg_amd_$l.so
  echo -n "$l: "; JPEG_AMD_LIBRARY=$lib python tools/run_c3.py 300 4096 4096 1 2>/dev/null
  echo -n "$l: "; JPEG_AMD_LIBRARY=$lib python tools/run_c3.py 100 2048 2048 16 2>/dev/null
done
